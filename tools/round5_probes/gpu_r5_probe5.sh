#!/bin/bash
# round 5: kernel H2's strip shape (huge_variant 5 = 64x16 strips, 7 = 64x32, both at 8 waves/SIMD) and workgroups per tile by record count
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]" | cut -c 1-120; }
S="1.25e8 ntotal=1e9 first=375000000 reorder=32"
run $S
for v in 5 7; do for sp in 64 96 128 192 256; do run $S huge_variant=$v huge_split=$sp; done; done
run 1e7 reorder=32
for v in 5 7; do for sp in 32 64 128 256; do run 1e7 reorder=32 huge_variant=$v huge_split=$sp; done; done
run 1e8 reorder=32
for sp in 128 192 256 384; do run 1e8 reorder=32 huge_variant=7 huge_split=$sp; done
run 1e9 reorder=32
for sp in 192 384; do run 1e9 reorder=32 huge_variant=7 huge_split=$sp; done
run 3e6 reorder=32
for v in 5 7; do for sp in 16 32 64 128; do run 3e6 reorder=32 huge_variant=$v huge_split=$sp; done; done
