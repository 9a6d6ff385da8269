#!/bin/bash
# round-6 quick timing set: 1e9 camera A, 1e8 camera-B-like zooms, 1e7 weighted, 5e7 rgb 2048^2 (extra args = options for every run)
python tools/gpu_bench_sweep.py 1e9 frames=4 "$@" 2>&1 | grep -E "frame 3"
python tools/gpu_bench_sweep.py 1e8 frames=4 reorder=8 scale=20 "$@" 2>&1 | grep -E "frame 3"
python tools/gpu_bench_sweep.py 1e8 frames=4 reorder=8 scale=50 "$@" 2>&1 | grep -E "frame 3"
python tools/gpu_bench_sweep.py 1e7 frames=4 reorder=8 mode=weighted "$@" 2>&1 | grep -E "frame 3"
python tools/gpu_bench_sweep.py 5e7 frames=4 reorder=8 mode=rgb R=2048 "$@" 2>&1 | grep -E "frame 3"
python tools/gpu_bench_sweep.py 1.25e8 frames=4 reorder=8 ntotal=1e9 first=3.75e8 "$@" 2>&1 | grep -E "frame 3"
