#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for c in 0 2048 4096; do run 1e9 reorder=32 frames=4 mid_item_records=$c; done
for c in 0 1024; do run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4 mid_item_records=$c; done
for sp in 128 192 384; do run 1e9 reorder=32 frames=4 huge_split=$sp; done
python tools/gpu_fuzz.py 7000 7300 weighted 2>&1 | tail -1
python tools/gpu_fuzz.py 7000 7150 depth 2>&1 | tail -1
python tools/gpu_fuzz.py 7000 7150 rgb 2>&1 | tail -1
