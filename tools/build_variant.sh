#!/bin/bash
# Builds an alternative libtopsy_splat (extra compiler defines) next to the product library, for A/B measurements:
#   tools/build_variant.sh <suffix> -DTSP_HDEAL=4 ...   ->  topsy_amd/libtopsy_splat_<suffix>.so  (select it with TOPSY_SPLAT_LIB)
set -e
SUF=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
B=$ROOT/topsy_amd/csrc/build_$SUF
mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -I$ROOT/include $@"
for f in tsp_api tsp_splat_generic tsp_pipeline tsp_gather tsp_colormap tsp_postpass tsp_data tsp_comm tsp_group; do
  /opt/rocm/bin/hipcc $FLAGS -c $ROOT/topsy_amd/csrc/$f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/topsy_amd/libtopsy_splat_$SUF.so $B/*.o -ldl
rm -rf $B
echo built topsy_amd/libtopsy_splat_$SUF.so
