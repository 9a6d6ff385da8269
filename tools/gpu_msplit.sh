#!/bin/bash
# kernel M's fixed cost: the h-capped snapshot (its only mid records are small footprints outside kernel S's window) and
# the headline at several splits
cd $GRAFT_REPO_ROOT
for ms in 4 16 32 64 128 256; do
  echo "=== hcap mid_split=$ms"; python3 tools/gpu_bench_sweep.py 1.25e8 hcap=8 reorder=50 frames=4 mid_split=$ms 2>&1 | grep "frame 3\|fragments"
done
for ms in 32 64 128 256; do
  echo "=== headline mid_split=$ms"; python3 tools/gpu_bench_sweep.py 1.25e8 reorder=50 frames=4 mid_split=$ms 2>&1 | grep "frame 3"
done
