#!/bin/bash
# packed FMA for the channel accumulators of kernels G / H2 (two and three channels)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
run 5e7 reorder=8 frames=4 mode=rgb R=2048
run 1e7 reorder=8 frames=4 mode=weighted
run 1e7 reorder=8 frames=4 mode=depth
run 1e8 reorder=8 frames=4 mode=weighted
run 1e9 reorder=32 frames=4
