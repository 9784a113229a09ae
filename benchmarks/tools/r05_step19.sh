#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for leg in ilrma_iss2 ilrma_ip1; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$leg -- python benchmarks/tools/leg_run.py $leg 32 20 > /dev/null 2>&1
f=$(find /tmp/prof_$leg -name "*kernel_stats.csv" | head -1)
echo "== $leg"; python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s %5s %10.1f %6s" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
done
