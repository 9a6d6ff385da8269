#!/bin/bash
# H2: record prefetch that is not waited for at once (A) against the conditional fetch (B), 8 and 7 waves per SIMD, several splits
S="1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4"
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for lib in "" topsy_amd/libtopsy_splat_condfetch.so; do
  export TOPSY_SPLAT_LIB=$lib; [ -z "$lib" ] && unset TOPSY_SPLAT_LIB
  echo "#### lib: ${lib:-product}"
  run $S; run $S huge_variant=6; run $S huge_split=192; run $S huge_split=256
  run 1e9 reorder=32 frames=4; run 1e9 reorder=32 frames=4 huge_variant=6; run 1e9 reorder=32 frames=4 huge_split=384
done
