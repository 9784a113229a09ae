#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_ipa; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q -k "ipa" > $out/tests.log 2>&1; tail -4 $out/tests.log
timeout 900 python bench.py --only-pairwise --no-cpu-baseline > $out/bench_pairwise.json 2> $out/bench.err; tail -3 $out/bench.err
SSSPY_AMD_IPA_PER_SOURCE=1 timeout 900 python bench.py --only-pairwise --no-cpu-baseline > $out/bench_pairwise_persource.json 2>> $out/bench.err
python - <<'P'
import json
for f in ("bench_pairwise.json","bench_pairwise_persource.json"):
    try:
        d=json.loads(open("gpurun_out/r05_ipa/"+f).read().strip().splitlines()[-1])["pairwise_ipa"]
    except Exception as e:
        print(f, "ERR", e); continue
    print(f)
    for k,v in d.items():
        if isinstance(v,dict): print(" ",k,{b:(x.get("ms_per_step"),x.get("frac"),x.get("error")) for b,x in v.items() if b.startswith("b")})
P
