#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths_agree.py -x -q -k "mnmf" 2>&1 | tail -3
timeout 200 python benchmarks/tools/mnmf_steps.py 32 2>&1 | tail -1
timeout 200 python benchmarks/tools/mnmf_steps.py 128 2>&1 | tail -1
timeout 300 python benchmarks/tools/call_timeline.py 100 2>&1 | grep -v "^$" | head -90
