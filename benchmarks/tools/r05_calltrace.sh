#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_call; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python benchmarks/tools/call_once.py 100 > $out/log 2>&1
grep __call__ $out/log
f=$(find $out/t -name '*kernel_stats.csv' | head -1)
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("   %-100s %5s x %8.1f us  %5.1f%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
P
cp "$f" $out/call_kernel_stats.csv; rm -rf $out/t
