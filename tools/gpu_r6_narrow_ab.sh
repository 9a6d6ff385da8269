#!/bin/bash
# kernel N A/B (round 6): mid footprints below mid_narrow_px through kernel N against everything through kernel G
mkdir -p gpurun_out/r6d
for px in 0 24000 32000 45254 64000; do
  echo "== mid_narrow_px_milli=$px, 1e9 camera A"; python tools/gpu_bench_sweep.py 1e9 frames=3 mid_narrow_px_milli=$px 2>&1 | grep -E "frame 2|fragments by"
done
for px in 0 32000 64000; do
  echo "== mid_narrow_px_milli=$px, 1e8 scale 20 / 50"; python tools/gpu_bench_sweep.py 1e8 frames=3 reorder=8 scale=20 mid_narrow_px_milli=$px 2>&1 | grep -E "frame 2"
  python tools/gpu_bench_sweep.py 1e8 frames=3 reorder=8 scale=50 mid_narrow_px_milli=$px 2>&1 | grep -E "frame 2"
  echo "== 1e7 weighted / 5e7 rgb 2048"; python tools/gpu_bench_sweep.py 1e7 frames=3 reorder=8 mode=weighted mid_narrow_px_milli=$px 2>&1 | grep -E "frame 2"
  python tools/gpu_bench_sweep.py 5e7 frames=3 reorder=8 mode=rgb R=2048 mid_narrow_px_milli=$px 2>&1 | grep -E "frame 2"
done
