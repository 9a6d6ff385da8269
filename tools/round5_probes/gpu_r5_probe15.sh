#!/bin/bash
# round 5: workgroups per tile of kernel M (mid_split) again, after its branch-free stepping
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]" | cut -c 1-100; }
S="1.25e8 ntotal=1e9 first=375000000"
for sp in 48 64 96 128 192; do run 1e9 reorder=32 mid_split=$sp; done
for sp in 32 48 64 96 128 192; do run $S reorder=8 mid_split=$sp; done
for sp in 32 64 128; do run 1e7 reorder=8 mid_split=$sp; done
for sp in 32 64 128; do run 5e7 reorder=8 mode=rgb R=2048 mid_split=$sp; done
