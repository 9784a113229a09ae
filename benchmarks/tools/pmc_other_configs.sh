#!/bin/bash
# HBM traffic (PMC) of the configs[2] / configs[3] kernels at 32 mixtures, as profile_round.sh takes it
# for the headline: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, no tracing.
#   gpurun: bash benchmarks/tools/pmc_other_configs.sh ; then python benchmarks/tools/pmc_other_configs.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/pmc_other
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -- \
    python benchmarks/other_configs.py --batch 32 --only iva_iss,fastmnmf --iters 3 > $out/$c.log 2>&1
done
find $out -name "*counter_collection.csv" | head
