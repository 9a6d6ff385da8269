#!/bin/bash
# Diagnostic PMC passes (SQ wait/active/LDS-conflict counters) over a short frame loop; run on the GPU box.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/diag
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
N=${1:-1.25e8}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o run -- python3 tools/gpu_bench_sweep.py $N frames=2 > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, collections, glob, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/diag/p*/run_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "splat_" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {sum(v)/len(v):.4g}")
PY
