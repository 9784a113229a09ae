#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export LEG_SOURCES=8
bash benchmarks/tools/r05_legprof.sh r05_legs8 16 ilrma_ip2 ilrma_ipa ilrma_ip1 2>&1 | cut -c1-150
