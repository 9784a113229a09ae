#!/bin/bash
cd $GRAFT_REPO_ROOT
SSSPY_AMD_MNMF_GLDS_TSTORE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fast_gauss_mnmf" 2>&1 | tail -3
for i in 1 2; do
SSSPY_AMD_MNMF_GLDS_TSTORE=1 timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1 | sed 's/^/TSTORE /'
timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1
done
SSSPY_AMD_MNMF_GLDS_TSTORE=1 timeout 200 python benchmarks/tools/mnmf_steps.py 128 2>&1 | tail -1 | sed 's/^/TSTORE /'
timeout 200 python benchmarks/tools/mnmf_steps.py 128 2>&1 | tail -1
