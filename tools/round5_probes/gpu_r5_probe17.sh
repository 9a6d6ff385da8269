#!/bin/bash
# strata count against kernel times: one shard of the 1e9 snapshot (index range 3 of 8) and the whole snapshot
for r in 1 2 3 4 6 8 16; do
  echo "== shard 3of8 strata $r"; python tools/gpu_bench_sweep.py 1.25e8 ntotal=1e9 first=3.75e8 reorder=$r frames=4 2>&1 | grep -E "frame [23]|reorder"
done
for r in 16 24 32 48; do
  echo "== whole 1e9 strata $r"; python tools/gpu_bench_sweep.py 1e9 reorder=$r frames=4 2>&1 | grep -E "frame [23]|reorder"
done
