#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "eigh or sqrtmh or gmeanmh or psd or hermitian or operators or linalg or gauss_mnmf or ipa or golden" 2>&1 | grep -v "^  \|Warning\|^tests/" | tail -8
