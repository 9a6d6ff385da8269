#!/bin/bash
# round 5: the H2 launch rule by record count (auto), strata counts and kernel S's grid bound by snapshot size
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]" | cut -c 1-120; }
S="1.25e8 ntotal=1e9 first=375000000"
for r in 32 8; do run $S reorder=$r; run 1e7 reorder=$r; run 1e8 reorder=$r; run 3e6 reorder=$r; run 1e6 reorder=$r; run 1e7 reorder=$r mode=weighted; done
run $S reorder=8 stream_blocks_per_cu=50
run $S reorder=8 stream_blocks_per_cu=25
run 1e9 reorder=32
run 1e9 reorder=32 stream_blocks_per_cu=50
run 1e9 reorder=64
run 1.25e8 reorder=32
run 1.25e8 reorder=8
run 5e7 reorder=32 mode=rgb R=2048
run 5e7 reorder=8 mode=rgb R=2048
