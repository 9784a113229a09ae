#!/bin/bash
# A / B of builds of the library on the bench line (and the ISS lines): 
#   benchmarks/tools/ab_lib.sh [-r rounds] <other .so> [<other .so> ...]       (GPU box, repo-relative paths)
# (variants: benchmarks/tools/build_variant.sh in the build container)
rounds=1
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
cd $GRAFT_REPO_ROOT
summ() { python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('%-28s value %8.0f  ms/step %.4f  kernels %s' % (sys.argv[1], b['value'], b['ms_per_step'], r['per_kernel_ms']))" $1; }
one() {  # <label> <lib or empty>
  SSSPY_AMD_LIB=${2:+$GRAFT_REPO_ROOT/$2} python bench.py --no-cpu-baseline --no-single --steps 20 --warmup 5 2>/dev/null | summ $1
  [ -n "$AB_ISS" ] && SSSPY_AMD_LIB=${2:+$GRAFT_REPO_ROOT/$2} python benchmarks/iva_lines.py 2>/dev/null | grep "128\|32 mix" | tr '\n' ' ' && echo
}
for i in $(seq $rounds); do
  one base ""
  for lib in "$@"; do one $(basename $lib .so | sed s/libssspy_amd_//) $lib; done
done
one base ""
