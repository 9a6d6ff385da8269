#!/bin/bash
# The HDEAL 4-vs-16 FETCH_SIZE experiment for kernel H2: build the variant first with
#   tools/build_variant.sh hdeal4 -DTSP_HDEAL=4
# then run this on the GPU box (it also runs the -m gpu suite and the bench + profiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r3a/pytest.log 2>&1
tail -5 gpurun_out/r3a/pytest.log
tools/profile_bench.sh > gpurun_out/r3a/profile.log 2>&1
tail -3 gpurun_out/r3a/profile.log
cd /tmp && export TMPDIR=/tmp
for v in hdeal4 ""; do
  if [ -n "$v" ]; then export TOPSY_SPLAT_LIB=$GRAFT_REPO_ROOT/topsy_amd/libtopsy_splat_$v.so; else unset TOPSY_SPLAT_LIB; fi
  O=$GRAFT_REPO_ROOT/gpurun_out/r3a/fetch_${v:-hdeal16}
  ( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O -o run -- python3 bench.py --headline-only --steps 3 --warmup 1 > $O.log 2>&1 )
  ( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d ${O}_tcc -o run -- python3 bench.py --headline-only --steps 3 --warmup 1 > ${O}_tcc.log 2>&1 )
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for p in sorted(glob.glob("gpurun_out/r3a/fetch_*/run_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "splat_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(p)
    for k, d in acc.items():
        print("  ", k, {c: sum(v) / len(v) for c, v in d.items()})
PY
