#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths_agree.py -x -q -k "ilrma" 2>&1 | tail -3
bash benchmarks/tools/ab_lib.sh -r 2 ssspy_amd/lib/libssspy_amd_noskip.so 2>&1 | tail -8
timeout 300 python benchmarks/cache_energy.py --seconds 4 > gpurun_out/r05_cache_energy.json 2>/dev/null; cat gpurun_out/r05_cache_energy.json | python -c "
import json,sys
d=json.load(sys.stdin); print('idle', d['idle'].get('socket_w'))
for l in d['legs']: print(l)
"
