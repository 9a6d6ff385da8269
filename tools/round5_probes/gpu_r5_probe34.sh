#!/bin/bash
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for lib in "" topsy_amd/libtopsy_splat_gocc8.so topsy_amd/libtopsy_splat_ghr16.so; do
  export TOPSY_SPLAT_LIB=$lib; [ -z "$lib" ] && unset TOPSY_SPLAT_LIB
  echo "#### lib: ${lib:-product}"
  run 1e9 reorder=32 frames=4
  run 1e7 reorder=8 frames=4 mode=weighted
  run 5e7 reorder=8 frames=4 mode=rgb R=2048
done
