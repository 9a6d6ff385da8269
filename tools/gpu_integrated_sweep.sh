#!/bin/bash
# kernel I: workgroups per band (mega_split) at integrated_px = 256 and 384 on the headline snapshot
cd $GRAFT_REPO_ROOT
for px in 256 384; do for sp in 2 4 8 16 32; do
  echo "=== integrated_px=$px mega_split=$sp"; python3 tools/gpu_bench_sweep.py 1.25e8 reorder=50 frames=4 integrated_px=$px mega_split=$sp 2>&1 | grep "frame 3"
done; done
