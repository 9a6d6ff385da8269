#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases: per-kernel time and PMC sums (kernel names truncated)."""
import sqlite3, sys, collections
def short(n):
    n = n.replace('void ', '')
    if 'rocprim' in n: return 'rocprim::' + ('radix_sort' if 'radix_sort' in n else 'other')
    return n.split('(')[0][:48]
for path in sys.argv[1:]:
    con = sqlite3.connect(path); cur = con.cursor()
    print("==", path)
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name order by 3 desc"))
    for name, n, tot, avg, mn in rows[:10]:
        print(f"  {short(name):50s} calls {n:4d} total {tot/1e6:10.3f} ms avg {avg/1e3:10.1f} us min {mn/1e3:10.1f} us")
    try:
        q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
        acc = collections.defaultdict(dict)
        for name, cname, val, cnt in cur.execute(q):
            acc[short(name)][cname] = (val, cnt)
        for k, d in acc.items():
            if 'splat' not in k and 'colormap' not in k: continue
            print("  PMC", k)
            for c, (v, cnt) in sorted(d.items()):
                print(f"      {c:28s} {v:16.4g}  (/dispatch {v/max(cnt,1):.4g} x{cnt})")
    except Exception as e:
        print("  (no pmc)", e)
