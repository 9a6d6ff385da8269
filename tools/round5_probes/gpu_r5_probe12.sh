#!/bin/bash
# round 5: kernel M, two-channel renders: one selecting loop (product) against the first branch-free iteration ("m1"), same box
cd $GRAFT_REPO_ROOT
for a in "1e7 reorder=8 mode=weighted" "1.25e8 reorder=8 mode=weighted" "1e8 reorder=8" "1e7 reorder=8 mode=depth"; do
  tools/gpu_ab.sh "- m1" $a 2>&1 | grep -E "===|frame [34]" | cut -c 1-110
done
