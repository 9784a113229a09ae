#!/bin/bash
# usage: benchmarks/pmc_single.sh <tag> <COUNTER> [<COUNTER> ...]   (one rocprofv3 --pmc pass, no tracing)
# Counter CSV of the single-mixture loop (benchmarks/single_mixture.py 12) under gpurun_out/<tag>/.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --output-format csv -d $out -- python benchmarks/single_mixture.py 12 > $out/run.log 2>&1 || tail -5 $out/run.log
python benchmarks/pmc_summary.py $out
