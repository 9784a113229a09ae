#!/usr/bin/env python3
"""Timeline of the single-mixture iteration from a rocprofv3 --kernel-trace CSV: per kernel of one
steady-state iteration the average duration and the gap to the previous kernel's end.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/<tag> -- python benchmarks/single_mixture.py 60
    python benchmarks/single_trace.py gpurun_out/<tag> [kernels-per-iteration-marker]
"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
paths = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
path = max(paths, key=os.path.getmtime)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the iteration = the span between consecutive launches of the first kernel of the last third
tail = rows[len(rows) // 2:]
first = tail[0]["Kernel_Name"]
starts = [i for i, r in enumerate(tail) if r["Kernel_Name"] == first]
period = starts[1] - starts[0]
its = [tail[s:s + period] for s in starts[:-1] if s + period <= len(tail)]
its = [it for it in its if [r["Kernel_Name"] for r in it] == [r["Kernel_Name"] for r in its[0]]]
dur = collections.defaultdict(float)
gap = collections.defaultdict(float)
for it_idx, it in enumerate(its):
    for k, r in enumerate(it):
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if k > 0:
            gap[k] += (int(r["Start_Timestamp"]) - int(it[k - 1]["End_Timestamp"])) / 1e3
n = len(its)
total = 0.0
print("{} iterations of {} kernels ({})".format(n, period, os.path.basename(path)))
for k, r in enumerate(its[0]):
    d, g = dur[k] / n, gap[k] / n
    total += d + g
    print("{:2d} {:60s} dur {:7.2f} us  gap {:6.2f} us  grid {} wg {}".format(
        k, r["Kernel_Name"][:60], d, g, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
span = (int(its[-1][-1]["End_Timestamp"]) - int(its[0][0]["Start_Timestamp"])) / 1e3 / n
print("sum dur+gap {:.2f} us; iteration period {:.2f} us".format(total, span))
