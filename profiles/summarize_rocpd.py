#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel stats / PMC counters) as text."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
con.row_factory = sqlite3.Row
print("== top_kernels ==")
try:
    for r in con.execute("select * from top_kernels"):
        print(dict(r))
except Exception as e:
    print("n/a", e)
print("== kernels (per dispatch) ==")
try:
    rows = list(con.execute("select name, start, end, (end-start) as dur_ns, grid_x as grid, workgroup_x as wg, "
                            "static_lds_size as lds, vgpr_count, sgpr_count, scratch_size from kernels order by start"))
    for r in rows:
        if "fiasco" in r["name"]:
            print(dict(r))
    byname = {}
    for r in rows:
        byname.setdefault(r["name"], []).append(r["dur_ns"])
    print("== per-kernel summary ==")
    for k, v in byname.items():
        print("%-40s calls %4d  total %.3f ms  avg %.3f ms  min %.3f ms  max %.3f ms" %
              (k[:40], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6))
except Exception as e:
    print("n/a", e)
print("== counters ==")
try:
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    agg = {}
    for r in con.execute("select * from counters_collection"):
        d = dict(r)
        key = (d.get("kernel_name") or d.get("name"), d.get("counter_name"))
        agg.setdefault(key, []).append(d.get("value"))
    for (k, c), v in agg.items():
        print("%-40s %-14s dispatches %4d  sum %.6g  avg %.6g" % (str(k)[:40], c, len(v), sum(v), sum(v) / len(v)))
except Exception as e:
    print("n/a", e)
