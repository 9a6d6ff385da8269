#!/usr/bin/env python3
"""Condense gpurun_out/bench_prof (tools/profile_bench.sh) into small tracked files under profiles/."""
import csv, collections, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/bench_prof"      # a third argument (any) keeps profiles/latest_bench_counters.json as it is
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.replace("void ", "")
    if "rocprim" in n:
        return "rocprim::" + ("radix_sort_onesweep" if "radix_sort" in n else "other")
    return n.split("(")[0]


def stats(trace_dir, out_name):
    p = f"{src}/{trace_dir}/run_kernel_stats.csv"
    if not os.path.exists(p):
        return
    rows = list(csv.DictReader(open(p)))
    with open(out_name, "w") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def counters(dirs):
    pmc = collections.defaultdict(dict)
    for d in dirs:
        p = f"{src}/{d}/run_counter_collection.csv"
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, dd in acc.items():
            for c, v in dd.items():
                pmc[k][c] = sum(v) / len(v)
    out = {}
    for k, d in pmc.items():
        if "tsp::" not in k:
            continue
        e = dict(d)
        if "FETCH_SIZE" in d: e["hbm_read_bytes_corrected"] = 2 * d["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in d: e["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in d: e["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0), 1)
        if "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"] > 0:
            e["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        out[k] = e
    return out


stats("trace", f"profiles/{tag}_bench_kernel_stats.csv")
stats("hcap_trace", f"profiles/{tag}_hcapped_kernel_stats.csv")
stats("shard_trace", f"profiles/{tag}_shard3of8_kernel_stats.csv")
out = {"tag": tag, "command": "python3 bench.py --headline-only under rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum "
                  "TCC_MISS_sum / --pmc SQ_INSTS_* / --pmc SQ_LDS_* SQ_WAIT_* (separate passes); `hcapped` = the same with --particles-per-gpu 1.25e8 "
                  "--h-cap-px 8; `shard3of8` = with --as-shard 8:3 (the index range [3.75e8, 5e8) of the 1e9 snapshot: what one of 8 GPUs renders)",
       "note": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads: "
               "HBM read bytes ~= 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section).  Every value is the mean over the launches of "
               "the pass; all launches belong to the headline workload (--headline-only)",
       "per_kernel": counters(("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds")),
       "hcapped_per_kernel": counters(("hcap_pmc_fetch", "hcap_pmc_lds")),
       "shard3of8_per_kernel": counters(("shard_pmc_fetch", "shard_pmc_write", "shard_pmc_sq"))}
try:
    out["bench_line"] = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
except Exception as ex:
    out["bench_line"] = str(ex)
json.dump(out, open(f"profiles/{tag}_bench_counters.json", "w"), indent=1)
if len(sys.argv) <= 3:
    shutil.copy(f"profiles/{tag}_bench_counters.json", "profiles/latest_bench_counters.json")
print(open(f"profiles/{tag}_bench_kernel_stats.csv").read()[:1500] if os.path.exists(f"profiles/{tag}_bench_kernel_stats.csv") else "")
print(json.dumps(out["per_kernel"], indent=1)[:3000])
print(json.dumps(out["hcapped_per_kernel"], indent=1)[:2000])
