#!/bin/bash
# waves (= work items) per workgroup of kernel G: 1 / 2 / 4 / 8 share one LUT copy
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for lib in "" topsy_amd/libtopsy_splat_gt128.so topsy_amd/libtopsy_splat_gt64.so topsy_amd/libtopsy_splat_gt512.so; do
  export TOPSY_SPLAT_LIB=$lib; [ -z "$lib" ] && unset TOPSY_SPLAT_LIB
  echo "#### lib: ${lib:-product (256 threads)}"
  run 1e9 reorder=32 frames=4
  run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4
  run 5e7 reorder=8 frames=4 mode=rgb R=2048
done
