#!/bin/bash
# rocprofv3 kernel statistics of separator legs: benchmarks/tools/r05_legprof.sh <tag> <batch> leg...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; B=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
for leg in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$leg -- python benchmarks/tools/leg_run.py $leg $B 10 > $out/$leg.log 2>&1
  grep "ms per iteration" $out/$leg.log
  f=$(find $out/$leg -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("   %-90s %4s x %9.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
P
  cp "$f" $out/${leg}_b${B}_kernel_stats.csv; rm -rf $out/$leg
done
