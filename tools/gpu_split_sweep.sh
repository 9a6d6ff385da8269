#!/bin/bash
# workgroups-per-tile options of kernels M / H2 and kernel S's grid bound around their defaults: N=1e9 tools/gpu_split_sweep.sh
cd $GRAFT_REPO_ROOT
N=${N:-1.25e8}
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep "frame [34]" | cut -c 1-90; }
run $N
for sp in 64 128 192 384 512; do run $N huge_split=$sp; done
for c in 256 512 1024 2048; do run $N mid_item_records=$c; done
for b in 50 150 200; do run $N stream_blocks_per_cu=$b; done
for v in 2 5 6 7; do run $N huge_variant=$v; done
