#!/bin/bash
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for p in 8000 11000 13000 16000; do run 1e9 reorder=32 frames=4 p_small_milli=$p; done
for p in 8000 12000 16000; do run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4 p_small_milli=$p; done
