#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ipa or IPA" 2>&1 | grep -v "^  \|Warning\|^tests/" | grep "Error\|assert\|kw\|E  \|passed\|failed" | head -12
