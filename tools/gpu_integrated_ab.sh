#!/bin/bash
# A/B of library builds with the option integrated_px: tools/gpu_integrated_ab.sh "<suffixes, - = product>"
cd $GRAFT_REPO_ROOT
for v in $1; do
  if [ "$v" = "-" ]; then unset TOPSY_SPLAT_LIB; else export TOPSY_SPLAT_LIB=$GRAFT_REPO_ROOT/topsy_amd/libtopsy_splat_$v.so; fi
  for px in 256 512; do echo "=== lib $v integrated_px=$px"; python3 tools/gpu_bench_sweep.py 1.25e8 reorder=50 frames=5 integrated_px=$px 2>&1 | grep "frame 4"; done
done
