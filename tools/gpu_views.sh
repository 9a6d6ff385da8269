#!/bin/bash
# other views of the 1.25e8-particle snapshot, exact path and with integrated_px = 256
cd $GRAFT_REPO_ROOT
for sc in 800 200 50 20; do for px in 0 256; do
  echo "=== scale=$sc integrated_px=$px"; python3 tools/gpu_bench_sweep.py 1.25e8 scale=$sc reorder=50 frames=4 integrated_px=$px 2>&1 | grep "frame 3\|fragments"
done; done
