#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "implied_filter or auxiva or iva" 2>&1 | grep -v "^  \|Warning\|^tests/" | tail -25
for leg in auxiva_iss2 auxiva_ipa; do
  for b in 32 1; do
    timeout 120 python benchmarks/tools/leg_run.py $leg $b 20
    SSSPY_AMD_NO_IMPLIED_FILTER=1 timeout 120 python benchmarks/tools/leg_run.py $leg $b 20 | sed 's/^/  on Y: /'
  done
done
