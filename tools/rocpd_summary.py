#!/usr/bin/env python3
"""Top kernels by total time from a rocprofv3 results database (rocpd sqlite): python tools/rocpd_summary.py x.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot, cnt = list(cur.execute(f"select sum(end-start)/1e3, count(*) from {kd}"))[0]
print(f"total kernel time {tot:.1f} us over {cnt} dispatches")
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit {n}")
for r in cur.execute(q):
    print("%-120s n=%7d total=%10.1fus avg=%8.1f max=%8.1f" % (r[0][:120], r[1], r[2], r[3], r[4]))
