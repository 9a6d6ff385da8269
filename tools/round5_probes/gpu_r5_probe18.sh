#!/bin/bash
# kernel S against the size of the index range drawn from the 1e9 snapshot, and against workgroups per CU at shard size
for n in 3.125e7 6.25e7 1.25e8 2.5e8 5e8; do
  echo "== range [0, $n) of 1e9"; python tools/gpu_bench_sweep.py $n ntotal=1e9 first=0 reorder=8 frames=4 2>&1 | grep -E "frame [23]"
done
for b in 2 4 8 16 32; do
  echo "== shard 3of8 stream_blocks_per_cu $b"; python tools/gpu_bench_sweep.py 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4 stream_blocks_per_cu=$b 2>&1 | grep -E "frame [23]"
done
