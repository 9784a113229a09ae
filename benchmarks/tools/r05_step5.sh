#!/bin/bash
cd $GRAFT_REPO_ROOT
SSSPY_AMD_MNMF_GLDS_PRIVATE_V=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fast_gauss_mnmf" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmark_sizes.py -x -q -k "fast_gauss_mnmf or configs3" 2>&1 | tail -3
for i in 1 2; do
SSSPY_AMD_MNMF_GLDS_PRIVATE_V=1 timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1 | sed 's/^/PRIV /'
timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1
done
SSSPY_AMD_MNMF_GLDS_PRIVATE_V=1 timeout 200 python benchmarks/tools/mnmf_steps.py 128 2>&1 | tail -1 | sed 's/^/PRIV /'
timeout 200 python benchmarks/tools/mnmf_steps.py 128 2>&1 | tail -1
timeout 300 python benchmarks/other_configs.py --only fastmnmf --batch 32 --iters 20 2>/dev/null | grep config
