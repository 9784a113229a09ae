#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q -k "ip2 or IP2 or pairwise or customfloor or fmnmf" 2>&1 | tail -3
LEG_SOURCES=8 timeout 200 python benchmarks/tools/leg_run.py ilrma_ip2 16 10 2>/dev/null | tail -1
LEG_SOURCES=6 timeout 200 python benchmarks/tools/leg_run.py ilrma_ip2 16 10 2>/dev/null | tail -1
LEG_SOURCES=8 timeout 200 python benchmarks/tools/leg_run.py auxiva_ip2 16 10 2>/dev/null | tail -1
