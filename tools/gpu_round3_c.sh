#!/bin/bash
cd $GRAFT_REPO_ROOT
for sp in 4 8 16 32 64 128; do echo "== mid_split $sp"; python3 tools/gpu_bench_sweep.py 1.25e8 hcap=8 frames=5 mid_split=$sp 2>&1 | grep "frame 4"; done
for sp in 32 64 128 256; do echo "== full mid_split $sp"; python3 tools/gpu_bench_sweep.py 1.25e8 frames=5 mid_split=$sp 2>&1 | grep "frame 4"; done
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $O/pmc -o run -- python3 bench.py --h-cap-px 8 --headline-only --steps 3 --warmup 1 > $O/pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/r3c/pmc/run_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "splat_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print("  ", k, {c: f"{sum(v) / len(v):.4g}" for c, v in d.items()})
PY
