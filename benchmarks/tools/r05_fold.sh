#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_ipa; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_paths_agree.py -x -q -k "ipa or iss2 or ISS2 or IPA or ilrma or folded" > $out/tests3.log 2>&1; tail -15 $out/tests3.log
for leg in ilrma_ipa ilrma_iss2 auxiva_ipa auxiva_iss2; do timeout 200 python benchmarks/tools/leg_run.py $leg 32 10 2>/dev/null | tail -1; done
