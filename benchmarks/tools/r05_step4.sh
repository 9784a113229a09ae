#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths_agree.py -x -q -k "loss or golden or deferred or determin or reproducible" 2>&1 | tail -3
timeout 300 python benchmarks/tools/call_timeline.py 100 2>&1 | grep -v "^$" | head -12
