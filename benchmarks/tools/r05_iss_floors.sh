#!/bin/bash
# Floors of the fused ISS kernel: build the variants first,
#   for v in 1 2 4 6; do benchmarks/tools/build_variant.sh issdbg$v "iss_fused=-DSSSPY_ISS_DBG=$v"; done
cd $GRAFT_REPO_ROOT
for v in "" issdbg1 issdbg2 issdbg4 issdbg6 ""; do
  echo "== ${v:-base}"
  SSSPY_AMD_LIB=${v:+$GRAFT_REPO_ROOT/ssspy_amd/lib/libssspy_amd_$v.so} timeout 300 python benchmarks/other_configs.py --batch 32 --only iva_iss --iters 10 2>&1 | grep -v amdgpu.ids | grep -i "ms\|iss" | head -4
done
