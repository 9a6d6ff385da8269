#!/bin/bash
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [23]"; }
for sp in 512 1024 2048; do run 1e9 reorder=32 frames=4 mid_split_gather=$sp; done
for sp in 128 256 512 1024; do run 1e9 reorder=32 frames=4 mid_split_gather=$sp debug_gather_order=1; done
for sp in 128 256 512; do run 1.25e8 ntotal=1e9 first=3.75e8 reorder=8 frames=4 mid_split_gather=$sp debug_gather_order=1; done
