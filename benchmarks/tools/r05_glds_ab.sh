#!/bin/bash
# Round 5, A / B of the LDS-DMA x passes of FastMNMF (run through gpurun from the repo root).
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r05_glds
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fast_gauss_mnmf" > $out/tests.log 2>&1
tail -5 $out/tests.log
for b in 1 32 128; do
  timeout 300 python benchmarks/other_configs.py --only fastmnmf --batch $b --iters 20 2>/dev/null | grep config >> $out/ab.txt
  SSSPY_AMD_MNMF_NO_GLDS=1 timeout 300 python benchmarks/other_configs.py --only fastmnmf --batch $b --iters 20 2>/dev/null | grep config | sed 's/^/NO_GLDS /' >> $out/ab.txt
done
cat $out/ab.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
  python benchmarks/other_configs.py --batch 32 --only fastmnmf --iters 10 > $out/stats.log 2>&1
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-200
