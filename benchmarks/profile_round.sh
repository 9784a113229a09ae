#!/bin/bash
# End-of-round measurement on the GPU box (run through gpurun from the repo root):
#   benchmarks/profile_round.sh <tag>
# Writes under gpurun_out/<tag>/: bench.json (default bench.py run), stats/ (rocprofv3
# --kernel-trace --stats of a short bench run), fetch/ and write/ (separate --pmc passes for the HBM
# traffic of the three pass kernels), other_configs.txt, other_stats/ (rocprofv3 statistics of
# configs[2] and configs[3] at 32 mixtures).  benchmarks/digest_profiles.py turns these
# into the tracked files under profiles/.
set -u
tag=${1:-r04}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- \
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- \
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single > $out/write.log 2>&1
python benchmarks/other_configs.py > $out/other_configs.txt 2>&1
python benchmarks/other_configs.py --batch 32 >> $out/other_configs.txt 2>&1
python benchmarks/other_configs.py --batch 128 --only fastmnmf --iters 10 >> $out/other_configs.txt 2>&1
python benchmarks/wide_mixtures.py >> $out/other_configs.txt 2>&1
python benchmarks/wide_basis.py 16 32 33 40 64 80 128 256 1024 >> $out/other_configs.txt 2>&1
python benchmarks/batch_sweep.py > $out/batch_sweep.txt 2>&1
python benchmarks/power_profile.py --seconds 5 > $out/power_profile.json 2> /dev/null
# per-kernel statistics of configs[2] / configs[3] at 32 mixtures
rocprofv3 --kernel-trace --stats --output-format csv -d $out/other_stats -- \
  python benchmarks/other_configs.py --batch 32 --only iva_iss,fastmnmf --iters 10 > $out/other_stats.log 2>&1
# single-mixture timelines (configs[1] / configs[3] literally) and the AuxIVA lines
python benchmarks/single_mixture.py 300 > $out/single.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/single_trace -- python benchmarks/single_mixture.py 60 > /dev/null 2>&1
python benchmarks/single_trace.py $out/single_trace >> $out/single.txt 2>&1
python benchmarks/single_mnmf.py 300 >> $out/single.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/single_mnmf_trace -- python benchmarks/single_mnmf.py 60 > /dev/null 2>&1
python benchmarks/single_trace.py $out/single_mnmf_trace >> $out/single.txt 2>&1
python benchmarks/iva_lines.py >> $out/single.txt 2>&1
tail -c 600 $out/bench.json
