#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
( python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "not config5 and not 100000000 and not overflow" ) > gpurun_out/r3d/pytest.log 2>&1
tail -3 gpurun_out/r3d/pytest.log
python3 tools/gpu_bench_sweep.py 1.25e8 hcap=8 frames=5 | grep -E "frame 4|frag"
python3 tools/gpu_bench_sweep.py 1.25e8 frames=5 | grep -E "frame 4|frag"
python3 tools/gpu_bench_sweep.py 1e9 reorder=400 frames=4 | grep -E "frame 3|frag"
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $O/pmc -o run -- python3 bench.py --h-cap-px 8 --headline-only --steps 3 --warmup 1 > $O/pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/r3d/pmc/run_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "splat_stream" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print("  ", k, {c: f"{sum(v) / len(v):.4g}" for c, v in d.items()})
PY
