#!/bin/bash
# kernel I on the other BASELINE configurations (exact path first, then integrated_px = 256 / 384 / 512)
cd $GRAFT_REPO_ROOT
for px in 0 256 384 512; do
  echo "=== config 2 (1e7 weighted) integrated_px=$px"; python3 tools/gpu_bench_sweep.py 1e7 mode=weighted reorder=16 frames=5 integrated_px=$px 2>&1 | grep "frame 4"
  echo "=== config 3 (1e8 density) integrated_px=$px"; python3 tools/gpu_bench_sweep.py 1e8 reorder=40 frames=4 integrated_px=$px 2>&1 | grep "frame 3"
done
for px in 0 256 512 1024; do
  echo "=== config 5 (5e7 rgb 2048^2) integrated_px=$px"; python3 tools/gpu_bench_sweep.py 5e7 mode=rgb R=2048 reorder=32 frames=4 integrated_px=$px 2>&1 | grep "frame 3"
done
for px in 0 256; do
  echo "=== 1e9 on one GPU integrated_px=$px"; python3 tools/gpu_bench_sweep.py 1e9 reorder=400 frames=3 integrated_px=$px 2>&1 | grep "frame 2"
done
