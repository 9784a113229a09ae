"""Average rocprofv3 --pmc counters per kernel: python benchmarks/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "ssspy" not in name and not name.startswith("k_"):
                continue
            short = name.split("(")[0].split("::")[-1]
            out[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())}
       for k, cs in out.items()}
print(json.dumps(res, indent=1))
