"""Kernel timeline of the last search step from a rocprofv3 results database (rocpd sqlite): name, duration, gap, start offset.
usage: rocpd_timeline.py results.db [marker-kernel-substring]  (the step = from the last launch of the marker kernel's predecessor)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(c.execute(f"select s.display_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
marker = sys.argv[2] if len(sys.argv) > 2 else "ivf_pair_query"
names = [r[0] for r in rows]
last = max(i for i, n in enumerate(names) if marker in n)
start = last
while start > 0 and "ivf_refine_finalize" not in names[start - 1]:
    start -= 1
prev, t0 = None, rows[start][1]
for n, s, e in rows[start:]:
    print(n[:64].ljust(64), "dur %8.1f us" % ((e - s) / 1e3), "gap %7.1f us" % (((s - prev) / 1e3) if prev else 0), "t %8.1f" % ((s - t0) / 1e3))
    prev = e
