#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "iss2 or ISS2 or pairwise or golden or implied or folded or floor" 2>&1 | grep -v "^  \|Warning\|^tests/" | tail -8
for leg in ilrma_iss2 auxiva_iss2; do
    timeout 120 python benchmarks/tools/leg_run.py $leg 32 20
    SSSPY_AMD_ISS2_ONE_LANE=1 timeout 120 python benchmarks/tools/leg_run.py $leg 32 20 | sed 's/^/  one lane: /'
done
LEG_SOURCES=8 timeout 120 python benchmarks/tools/leg_run.py ilrma_iss2 16 10
LEG_SOURCES=8 SSSPY_AMD_ISS2_ONE_LANE=1 timeout 120 python benchmarks/tools/leg_run.py ilrma_iss2 16 10 | sed 's/^/  one lane: /'
LEG_SOURCES=8 timeout 120 python benchmarks/tools/leg_run.py auxiva_iss2 16 10
LEG_SOURCES=8 SSSPY_AMD_ISS2_ONE_LANE=1 timeout 120 python benchmarks/tools/leg_run.py auxiva_iss2 16 10 | sed 's/^/  one lane: /'
