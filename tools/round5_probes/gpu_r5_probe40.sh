#!/bin/bash
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for lib in "" topsy_amd/libtopsy_splat_fold2k.so; do
  export TOPSY_SPLAT_LIB=$lib; [ -z "$lib" ] && unset TOPSY_SPLAT_LIB
  echo "#### lib: ${lib:-product}"
  run 5e7 reorder=8 frames=4 mode=rgb R=2048
  run 1e9 reorder=32 frames=4
  run 1e7 reorder=8 frames=4 mode=weighted
done
export TOPSY_SPLAT_LIB=topsy_amd/libtopsy_splat_fold2k.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
