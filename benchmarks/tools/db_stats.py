#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results .db (rocpd): `db_stats.py file.db [csv-out]`."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute("select %s, count(*), avg(end - start), min(end - start), max(end - start), "
                 "sum(end - start) from kernels group by %s order by 6 desc" % (name, name)).fetchall()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
if out:
    out.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n')
for n, calls, avg, mn, mx, tot in rows:
    if out:
        out.write('"%s",%d,%d,%.1f,%d,%d\n' % (n, calls, tot, avg, mn, mx))
    print("%-90s %5d %9.1f us" % (n[:90], calls, avg / 1e3))
