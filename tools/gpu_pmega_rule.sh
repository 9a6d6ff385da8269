#!/bin/bash
# Kernel H3 (matrix cores) on / off by snapshot size, zoom and mode: the sweep behind p_mega_px = p_mega2_px = 0 (end of round 4).
#   SIZES="1e7 1e8" OPTS="p_mega_px=768 p_mega_px=1536" EXTRA="scale=50" tools/gpu_pmega_rule.sh
for n in ${SIZES:-1e7 2e7 3e7 4e7 6e7 1e8}; do
  for o in "" ${OPTS:-p_mega_px=768 p_mega2_px=384}; do
    echo "== n=$n $EXTRA $o"; python3 tools/gpu_bench_sweep.py $n frames=6 $EXTRA $o | grep "frame [345]"
  done
done
