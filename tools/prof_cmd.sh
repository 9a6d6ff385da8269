#!/bin/bash
# Diagnostic PMC passes over an arbitrary python tool: tools/prof_cmd.sh <kernel-substring> tools/x.py args...   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/diag
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
PAT=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o run -- python3 "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
PAT=$PAT python3 - <<'PY'
import csv, collections, glob, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/diag/p*/run_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if os.environ["PAT"] not in k: continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for p in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/diag/p0/run_kernel_trace.csv")):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if os.environ["PAT"] not in k: continue
        dur[k.split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in acc.items():
    print(k, "launches", len(dur[k]), "ms", " ".join(f"{x:.3f}" for x in dur[k][:12]))
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):.4g}  last {v[-1]:.4g}")
PY
