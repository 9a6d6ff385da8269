#!/bin/bash
# per-kernel durations of one gpu_bench_sweep.py run (rocprofv3 --kernel-trace --stats); usage: tools/gpu_kernel_stats.sh <name> <sweep args>
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 tools/gpu_bench_sweep.py "$@" > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
for p in glob.glob("$OUT/trace/**/run_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:14]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
