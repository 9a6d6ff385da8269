#!/bin/bash
# H2 on 128 x 32 strips (two pixel columns per lane: 40 % fewer (footprint, strip) pairs) at 5 / 4 waves per SIMD
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fold_their" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for v in 1 8 9; do run 1e9 reorder=32 frames=4 huge_variant=$v; done
for v in 8 9; do for sp in 256 512; do run 1e9 reorder=32 frames=4 huge_variant=$v huge_split=$sp; done; done
for v in 1 8 9; do run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4 huge_variant=$v; done
