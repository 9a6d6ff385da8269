#!/bin/bash
# round 5: kernel H2 with the huge records binned by 64-row image band (huge_band_mib=0: one list for every tile, as before)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python3 tools/gpu_bench_sweep.py "$@" 2>&1 | grep -E "frame [34]" | cut -c 1-120; }
S="1.25e8 ntotal=1e9 first=375000000"
for hb in 6144 0; do
run 1e9 reorder=32 huge_band_mib=$hb
run $S reorder=32 huge_band_mib=$hb
run 1e8 reorder=32 huge_band_mib=$hb
run 1e7 reorder=32 huge_band_mib=$hb
run 1e7 reorder=32 mode=weighted huge_band_mib=$hb
run 5e7 reorder=32 mode=rgb R=2048 huge_band_mib=$hb
done
for sp in 128 192 256 384 512; do run 1e9 reorder=32 huge_split=$sp; done
for sp in 64 96 128 192 256; do run $S reorder=32 huge_variant=7 huge_split=$sp; done
for sp in 96 128 192 256 384; do run 1e8 reorder=32 huge_variant=7 huge_split=$sp; done
for sp in 64 128 256; do run 1e7 reorder=32 huge_variant=7 huge_split=$sp; done
for sp in 64 128 192 256; do run 1e7 reorder=32 mode=weighted huge_split=$sp; done
for sp in 16 32 64 128; do run 5e7 reorder=32 mode=rgb R=2048 huge_split=$sp; done
for g in 6144 0; do
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/r5_hb_fetch_$g -o run -- python3 tools/gpu_bench_sweep.py 1e9 reorder=32 frames=3 huge_band_mib=$g > gpurun_out/r5_hb_fetch_$g.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in glob.glob("gpurun_out/r5_hb_fetch_$g/**/run_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "splat_" in r["Kernel_Name"] or "band" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("huge_band_mib=$g", k, "FETCH_SIZE KiB mean", sum(v) / len(v), "-> HBM read GB", 2 * 1024 * sum(v) / len(v) / 1e9)
PY
done
# in-block arrangement of the load-time order: 0 = Morton, 1 = 64 x 8 transposition, 2 = by descending smoothing length
for il in 0 1 2; do
run 1e9 reorder=32 reorder_interleave=$il
run $S reorder=32 reorder_interleave=$il
run 1e8 reorder=32 reorder_interleave=$il
run 1.25e8 reorder=32 hcap=8 reorder_interleave=$il
run 1e7 reorder=32 mode=weighted reorder_interleave=$il
run 5e7 reorder=32 mode=rgb R=2048 reorder_interleave=$il
done
