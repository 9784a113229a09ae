#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q -k "ipa or IPA or lqpqm" 2>&1 | tail -4
for leg in ilrma_ipa auxiva_ipa; do timeout 200 python benchmarks/tools/leg_run.py $leg 32 10 2>/dev/null | tail -1; done
LEG_SOURCES=8 timeout 300 python benchmarks/tools/leg_run.py ilrma_ipa 16 5 2>/dev/null | tail -1
LEG_SOURCES=6 timeout 300 python benchmarks/tools/leg_run.py ilrma_ipa 16 5 2>/dev/null | tail -1
