#!/bin/bash
# round 5: kernel H2 with the band bins in place: records per dealing run (HDEAL 8 / 16 / 32) and footprints per float32 accumulator (512 / 1024)
cd $GRAFT_REPO_ROOT
S="1.25e8 ntotal=1e9 first=375000000"
for a in "1e9 reorder=32" "$S reorder=8" "1e8 reorder=8" "5e7 reorder=8 mode=rgb R=2048"; do
  tools/gpu_ab.sh "- hdeal8 hdeal32 fold1024" $a 2>&1 | grep -E "===|frame [34]" | cut -c 1-110
done
