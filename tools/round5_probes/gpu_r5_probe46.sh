#!/bin/bash
# H2 workgroups per tile for short record lists (9.4e4 records at 1e6 particles, 2.4e4 at 1e5, 3.4e5 at 1e7)
run() { echo "== $*"; python tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [3]"; }
for sp in 0 8 16 32 64 128; do run 1e6 reorder=8 frames=4 huge_split=$sp; done
for sp in 0 8 16 32 64; do run 1e5 reorder=8 frames=4 huge_split=$sp; done
for sp in 0 32 64 128 256; do run 1e7 reorder=8 frames=4 huge_split=$sp; done
for v in 5 7; do run 1e6 reorder=8 frames=4 huge_variant=$v; done
