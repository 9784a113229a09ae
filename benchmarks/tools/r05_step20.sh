#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" issdbg1 issdbg2 issdbg4 issdbg6 ""; do
  echo "== ${v:-base}"
  SSSPY_AMD_LIB=${v:+$GRAFT_REPO_ROOT/ssspy_amd/lib/libssspy_amd_$v.so} timeout 300 python benchmarks/other_configs.py --batch 32 --only iva_iss --iters 10 2>&1 | grep -v amdgpu.ids | grep -i "ms\|iss" | head -4
done
