#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for c in 256 512 1024 2048 4096; do run 1e9 reorder=32 frames=4 mid_item_records=$c; done
for c in 128 256 512 1024; do run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4 mid_item_records=$c; done
for c in 64 128 256 512; do run 1e7 reorder=8 frames=4 mid_item_records=$c; done
for c in 64 128 256; do run 1e6 reorder=8 frames=4 mid_item_records=$c; done
for c in 64 128; do run 1e5 reorder=8 frames=4 mid_item_records=$c; done
