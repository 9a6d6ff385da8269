#!/bin/bash
# round 5: kernel M's branch-free 8x8 stepping against the masked form (A/B builds, same box), all defaults of the round in place
cd $GRAFT_REPO_ROOT
S="1.25e8 ntotal=1e9 first=375000000"
for a in "1e9 reorder=32" "$S reorder=8" "1e8 reorder=8" "1e7 reorder=8" "1e7 reorder=8 mode=weighted" "5e7 reorder=8 mode=rgb R=2048" "1.25e8 reorder=8 hcap=8" "1.25e8 reorder=8 scale=50"; do
  tools/gpu_ab.sh "- mbranch" $a 2>&1 | grep -E "===|frame [34]" | cut -c 1-110
done
