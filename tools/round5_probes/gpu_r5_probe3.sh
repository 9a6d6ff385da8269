#!/bin/bash
# round 5, third probe: why the XCD-aware workgroup order is slower -- tiles per slice group (1 = the old tile-major order) for
# kernels H2 and M separately, and the fabric traffic with and without it
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]" | cut -c 1-150; }
S="1.25e8 ntotal=1e9 first=375000000 reorder=32"
for g in 0 1 2 4 8 16 32 64 256; do run $S xcd_group=$g xcd_group_mid=0; done
for g in 1 2 4 8 16 32 64 512; do run $S xcd_group=0 xcd_group_mid=$g; done
for g in 0 4 16; do run 1e9 reorder=32 xcd_group=$g xcd_group_mid=0; done
for g in 4 16; do run 1e9 reorder=32 xcd_group=0 xcd_group_mid=$g; done
for g in 0 16; do
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/r5_fetch_$g -o run -- python3 tools/gpu_bench_sweep.py 1.25e8 ntotal=1e9 first=375000000 reorder=32 frames=3 xcd_group=$g xcd_group_mid=$g > gpurun_out/r5_fetch_$g.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in glob.glob("gpurun_out/r5_fetch_$g/**/run_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "splat_" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("xcd_group=$g", k, "FETCH_SIZE KiB mean", sum(v) / len(v), "-> HBM read GB", 2 * 1024 * sum(v) / len(v) / 1e9)
PY
done
