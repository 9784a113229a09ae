#!/bin/bash
# A / B of two builds of the library on the bench line: benchmarks/tools/ab_lib.sh <other .so> [rounds]
# (build the variant with SSSPY_AMD_EXTRA_CXXFLAGS=..., copy the .so aside, rebuild the default)
other=$1; rounds=${2:-2}
cd $GRAFT_REPO_ROOT
summ() { python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('%-8s value %8.0f  ms/step %.4f  kernels %s' % (sys.argv[1], b['value'], b['ms_per_step'], r['per_kernel_ms']))" $1; }
for i in $(seq $rounds); do
  python bench.py --no-cpu-baseline --no-single --steps 20 --warmup 5 2>/dev/null | summ base
  SSSPY_AMD_LIB=$GRAFT_REPO_ROOT/$other python bench.py --no-cpu-baseline --no-single --steps 20 --warmup 5 2>/dev/null | summ other
done
