#!/bin/bash
# round 5, fourth probe: kernel M's padded LUT rows and kernel H2's no-change fast path (A/B builds), H2's strip shape / split at shard size
cd $GRAFT_REPO_ROOT
S="1.25e8 ntotal=1e9 first=375000000 reorder=32"
tools/gpu_ab.sh "- nofast nopad" 1e9 reorder=32 2>&1 | grep -E "===|frame [34]" | cut -c 1-100
tools/gpu_ab.sh "- nofast nopad" $S 2>&1 | grep -E "===|frame [34]" | cut -c 1-100
tools/gpu_ab.sh "- nofast nopad" 1.25e8 reorder=32 2>&1 | grep -E "===|frame [34]" | cut -c 1-100
tools/gpu_ab.sh "- nofast nopad" 5e7 reorder=32 mode=rgb R=2048 2>&1 | grep -E "===|frame [34]" | cut -c 1-100
tools/gpu_ab.sh "- nofast nopad" 1e7 reorder=32 mode=weighted 2>&1 | grep -E "===|frame [34]" | cut -c 1-100
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]" | cut -c 1-100; }
for sp in 96 128 160 192 256; do run $S huge_variant=7 huge_split=$sp; done
for sp in 64 96 128; do run $S huge_variant=5 huge_split=$sp; done
for sp in 64 128 192 256 384; do run 1e7 reorder=32 huge_split=$sp; done
for sp in 128 256 384; do run 1e7 reorder=32 huge_variant=7 huge_split=$sp; done
