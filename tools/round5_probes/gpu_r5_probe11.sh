#!/bin/bash
# round 5: kernel M's second branch-free iteration (per-block weights, padded tile, select only in the last column step) against the first ("m1"), same box
cd $GRAFT_REPO_ROOT
S="1.25e8 ntotal=1e9 first=375000000"
for a in "1e9 reorder=32" "$S reorder=8" "1e8 reorder=8" "1e7 reorder=8" "1e7 reorder=8 mode=weighted" "1.25e8 reorder=8 hcap=8" "1.25e8 reorder=8 scale=50" "1.25e8 reorder=8 mode=weighted"; do
  tools/gpu_ab.sh "- m1" $a 2>&1 | grep -E "===|frame [34]" | cut -c 1-110
done
