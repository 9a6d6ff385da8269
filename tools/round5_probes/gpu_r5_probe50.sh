#!/bin/bash
# fold interval of the float32 accumulators: 2048 (product) / 4096 / 8192 -- time, largest relative error, parity
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [3]"; }
for lib in "" topsy_amd/libtopsy_splat_fold4k.so topsy_amd/libtopsy_splat_fold8k.so; do
  export TOPSY_SPLAT_LIB=$lib; [ -z "$lib" ] && unset TOPSY_SPLAT_LIB
  echo "#### lib: ${lib:-product (2048)}"
  run 5e7 reorder=8 frames=4 mode=rgb R=2048
  run 1e9 reorder=32 frames=4
  python tools/gpu_accuracy.py 4e7 1e9 2>&1 | grep "default"
done
export TOPSY_SPLAT_LIB=topsy_amd/libtopsy_splat_fold8k.so
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -4
