#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
run 1e9 reorder=32 frames=4
run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4
run 1e7 reorder=8 frames=4 mode=weighted
run 5e7 reorder=8 frames=4 mode=rgb R=2048
run 1e6 reorder=8 frames=4
