#!/bin/bash
# A/B timing of library builds on the GPU box: tools/gpu_ab.sh "<lib suffixes, '-' = product>" <sweep args...>
# e.g. tools/gpu_ab.sh "base -" 1.25e8 hcap=8   (libtopsy_splat_base.so vs libtopsy_splat.so)
cd $GRAFT_REPO_ROOT
LIBS="$1"; shift
for v in $LIBS; do
  if [ "$v" = "-" ]; then unset TOPSY_SPLAT_LIB; else export TOPSY_SPLAT_LIB=$GRAFT_REPO_ROOT/topsy_amd/libtopsy_splat_$v.so; fi
  echo "=== lib ${v} : $@"
  python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -v "^frame [01]:"
done
