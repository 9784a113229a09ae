#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mnmf and (golden or lds_dma or handover)" 2>&1 | tail -2
timeout 900 python benchmarks/subbatch_sweep.py 2>/dev/null | tee gpurun_out/r05_subbatch_sweep.txt
