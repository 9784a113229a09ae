#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q -k "iva or IVA or aux or ipa" 2>&1 | tail -3
for leg in auxiva_iss2 auxiva_ipa; do for b in 32 128; do timeout 200 python benchmarks/tools/leg_run.py $leg $b 10 2>/dev/null | tail -1; done; done
