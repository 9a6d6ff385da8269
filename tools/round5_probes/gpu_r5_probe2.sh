#!/bin/bash
# round 5, second probe: kernel S with precomputed weights / deferred reservations (always on), lane decorrelation at reorder
# (reorder_interleave), XCD-aware workgroup order of the tile kernels (xcd_group), strata counts and class boundary at shard size
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]|fragments " | cut -c 1-150; }
run 1e9 reorder=32
run 1e9 reorder=32 reorder_interleave=0
run 1e9 reorder=32 xcd_group=0
run 1e9 reorder=32 debug_no_raster=1
run 1e9 reorder=32 p_small_milli=11300
run 1e9 reorder=32 p_small_milli=8000
S="1.25e8 ntotal=1e9 first=375000000"
run $S reorder=32
run $S reorder=32 reorder_interleave=0
run $S reorder=32 xcd_group=0
run $S reorder=16
run $S reorder=8
run $S reorder=4
for sp in 96 128 192; do run $S reorder=32 huge_variant=7 huge_split=$sp; done
for sp in 192 384; do run $S reorder=32 huge_variant=5 huge_split=$sp; done
run 1.25e8 reorder=32
run 1.25e8 reorder=32 huge_split=192
run 1e7 reorder=32 mode=weighted
run 5e7 reorder=32 mode=rgb R=2048
run 5e7 reorder=32 mode=rgb R=2048 xcd_group=0
