#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_paths_agree.py -x -q -k "iss or ISS or ilrma or folded" 2>&1 | tail -3
for b in 1 32 128; do
  timeout 200 python benchmarks/tools/leg_run.py ilrma_iss1 $b 10 2>/dev/null | tail -1 | sed 's/^/statistics: /'
  SSSPY_AMD_ISS1_STATISTICS=0 timeout 200 python benchmarks/tools/leg_run.py ilrma_iss1 $b 10 2>/dev/null | tail -1 | sed 's/^/fused sweep: /'
done
