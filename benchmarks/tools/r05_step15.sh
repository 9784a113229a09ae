#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q -k "mnmf" 2>&1 | tail -3
for b in 32 128; do timeout 200 python benchmarks/tools/leg_run.py fmnmf_ip2 $b 10 2>/dev/null | tail -1; done
