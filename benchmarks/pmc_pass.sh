#!/bin/bash
# usage: benchmarks/pmc_pass.sh <tag> <batch> <COUNTER> [<COUNTER> ...]   (one rocprofv3 --pmc pass, no tracing)
# Writes gpurun_out/<tag>/ with the counter CSV of `bench.py --batch <batch> --steps 2 --warmup 1`
# and prints the per-kernel averages (benchmarks/pmc_summary.py).
set -e
tag=$1; shift
batch=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --output-format csv -d $out -- python bench.py --batch $batch --steps 2 --warmup 1 --no-cpu-baseline --no-single > $out/run.log 2>&1 || tail -5 $out/run.log
python benchmarks/pmc_summary.py $out
