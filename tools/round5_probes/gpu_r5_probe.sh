#!/bin/bash
# round 5: where the whole 1e9 frame and one real index-range shard of it spend their time (kernel S with / without its rasteriser,
# workgroups per tile of kernels M / H2 at shard size)
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]|fragments" | cut -c 1-150; }
run 1e9 reorder=32
run 1e9 reorder=32 debug_no_raster=1
S="1.25e8 ntotal=1e9 first=375000000 reorder=32"
run $S
run $S debug_no_raster=1
for sp in 32 64 256; do run $S huge_split=$sp; done
for v in 7 5; do run $S huge_variant=$v; done
for v in 7; do for sp in 64 128; do run $S huge_variant=$v huge_split=$sp; done; done
for sp in 32 64 256; do run $S mid_split=$sp; done
for b in 25 50 200; do run $S stream_blocks_per_cu=$b; done
