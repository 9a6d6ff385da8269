#!/bin/bash
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for lib in "" topsy_amd/libtopsy_splat_mold.so topsy_amd/libtopsy_splat_mbperm.so; do
  export TOPSY_SPLAT_LIB=$lib; [ -z "$lib" ] && unset TOPSY_SPLAT_LIB
  echo "#### lib: ${lib:-product (rows by quad DPP, columns per step)}"
  run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4
  run 1e9 reorder=32 frames=4
  run 1e8 reorder=8 frames=4
done
