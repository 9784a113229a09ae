#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "ip1 or IP1 or operators or auxiva or iva or golden" 2>&1 | grep -v "^  \|Warning\|^tests/" | tail -6
LEG_SOURCES=8 timeout 120 python benchmarks/tools/leg_run.py auxiva_ip1 16 10
LEG_SOURCES=6 timeout 120 python benchmarks/tools/leg_run.py auxiva_ip1 16 10
