#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in "" dbg1 dbg2 dbg3; do
  SSSPY_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/ssspy_amd/lib/libssspy_amd_$lib.so} timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1
done
SSSPY_AMD_MNMF_NO_GLDS=1 timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_dma" 2>&1 | tail -2
