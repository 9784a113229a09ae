#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "folded_power or implied_filter" 2>&1 | tail -4
export FUZZ_KINDS=gauss,gauss_p1 FUZZ_ALGOS=ISS,ISS2,IPA FUZZ_MAX_SOURCES=4 FUZZ_ITER=8
timeout 900 python benchmarks/fuzz_parity.py 150 23 2>&1 | grep -v amdgpu.ids | grep -v Warn | tail -6
