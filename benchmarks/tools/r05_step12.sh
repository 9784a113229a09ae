#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "flooring or floor or fast_gauss_mnmf or rng_drawn" 2>&1 | tail -12
