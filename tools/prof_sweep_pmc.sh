#!/bin/bash
# PMC passes of tools/gpu_bench_sweep.py (any size / options), per splat kernel; run on the GPU box.
# Usage: tools/prof_sweep_pmc.sh <outdir-name> <gpu_bench_sweep.py arguments>
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o run -- python3 tools/gpu_bench_sweep.py "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_lds -o run -- python3 tools/gpu_bench_sweep.py "$@" > $OUT/pmc_lds.log 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("pmc_sq", "pmc_lds"):
    for p in glob.glob("$OUT/%s/**/run_counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "tsp::splat" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in acc.items():
            print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in dd.items()})
PY
