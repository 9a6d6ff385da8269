#!/usr/bin/env python3
"""Condense gpurun_out/bench_prof (tools/profile_bench.sh) into small tracked files under profiles/."""
import csv, collections, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
src = "gpurun_out/bench_prof"
os.makedirs("profiles", exist_ok=True)
def short(n):
    n = n.replace("void ", "")
    if "rocprim" in n:
        return "rocprim::" + ("radix_sort_onesweep" if "radix_sort" in n else "other")
    return n.split("(")[0]
rows = list(csv.DictReader(open(f"{src}/trace/run_kernel_stats.csv")))
with open(f"profiles/{tag}_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
pmc = collections.defaultdict(dict)
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    p = f"{src}/{d}/run_counter_collection.csv"
    if not os.path.exists(p): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in acc.items():
        for c, v in dd.items():
            pmc[k][c] = sum(v) / len(v)
out = {"command": "python3 bench.py (defaults) under rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes)",
       "note": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads: HBM read bytes ~= 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section)",
       "per_kernel": {}}
for k, d in pmc.items():
    if "tsp::" not in k: continue
    e = dict(d)
    if "FETCH_SIZE" in d: e["hbm_read_bytes_corrected"] = 2 * d["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in d: e["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in d: e["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0), 1)
    out["per_kernel"][k] = e
try:
    out["bench_line"] = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
except Exception as ex:
    out["bench_line"] = str(ex)
json.dump(out, open(f"profiles/{tag}_bench_counters.json", "w"), indent=1)
print(open(f"profiles/{tag}_bench_kernel_stats.csv").read()[:1500])
print(json.dumps(out["per_kernel"], indent=1)[:2500])
