#!/bin/bash
cd $GRAFT_REPO_ROOT
GM_WARM=60 timeout 900 python benchmarks/gmnmf_channels.py 8 4 8 2>&1 | grep channels | cut -c1-330
GM_WARM=60 SSSPY_AMD_GMNMF_SU_ROWS=0 timeout 900 python benchmarks/gmnmf_channels.py 8 8 2>&1 | grep channels | cut -c1-200 | sed 's/^/lane per matrix: /'
