#!/bin/bash
# End-of-round measurement on the GPU box (run through gpurun from the repo root):
#   benchmarks/profile_round.sh <tag>
# Writes under gpurun_out/<tag>/: bench.json (default bench.py run), stats/ (rocprofv3
# --kernel-trace --stats of a short bench run), fetch/ and write/ (separate --pmc passes for the HBM
# traffic of the three pass kernels), other_configs.txt, other_stats/ (rocprofv3 statistics of
# configs[2] and configs[3] at 32 mixtures).  benchmarks/digest_profiles.py turns these
# into the tracked files under profiles/.
set -u
tag=${1:-r06}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out $out/legs
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
# ---- round 5: per-step times of configs[3], the pairwise / IPA legs, the __call__ timeline, the
# energy per byte of the Infinity Cache, the headline in cache-sized sub-batches
python benchmarks/tools/mnmf_steps.py 32 > $out/mnmf_steps.txt 2>/dev/null
python benchmarks/tools/mnmf_steps.py 128 >> $out/mnmf_steps.txt 2>/dev/null
for leg in ilrma_ip2 ilrma_iss1 ilrma_iss2 ilrma_ipa auxiva_ip2 auxiva_iss2 auxiva_ipa fmnmf_ip2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/legs/$leg -- python benchmarks/tools/leg_run.py $leg 32 10 > $out/legs/$leg.log 2>&1
  f=$(find $out/legs/$leg -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/legs/${leg}_b32_kernel_stats.csv
  rm -rf $out/legs/$leg
done
# the same legs at 8 sources, 16 mixtures (the source counts above the tuned kernels)
for leg in ilrma_ip1 ilrma_ip2 ilrma_iss2 ilrma_ipa auxiva_iss2; do
  LEG_SOURCES=8 rocprofv3 --kernel-trace --stats --output-format csv -d $out/legs/$leg -- python benchmarks/tools/leg_run.py $leg 16 5 > $out/legs/${leg}_n8.log 2>&1
  f=$(find $out/legs/$leg -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/legs/${leg}_n8_b16_kernel_stats.csv
  rm -rf $out/legs/$leg
done
grep -h "ms per iteration" $out/legs/*.log > $out/legs_ms.txt
python benchmarks/tools/call_timeline.py 100 2>/dev/null | grep -v "^$" | head -12 > $out/call_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/call_stats -- python benchmarks/tools/call_once.py 100 >> $out/call_timeline.txt 2>&1
python benchmarks/cache_energy.py --seconds 4 > $out/cache_energy.json 2>/dev/null
# ---- round 6: FastGaussMNMF above 4 channels (iteration, Wiener filter, kernel statistics), the leg
# survey at 4 / 5 / 6 / 7 / 8 sources, GaussMNMF per channel count, the phases of one call of
# configs[2] / configs[3]
for m in 5 6 7 8; do python benchmarks/tools/fmnmf_wide.py $m; done > $out/fmnmf_wide.txt 2>/dev/null
for m in 6 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/fmw$m -- python benchmarks/tools/fmnmf_wide.py $m > /dev/null 2>&1
  f=$(find $out/fmw$m -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/fmnmf_wide_m${m}_kernel_stats.csv
  rm -rf $out/fmw$m
done
for n in 2 4 5 6 7 8; do python benchmarks/tools/leg_survey.py $n; done > $out/leg_survey.txt 2>/dev/null
python benchmarks/gmnmf_channels.py 8 4 5 6 7 8 > $out/gmnmf_channels.txt 2>/dev/null
for k in ilrma auxiva fmnmf; do python benchmarks/tools/call_any.py $k 100; done > $out/call_phases.txt 2>/dev/null
python benchmarks/tools/call_phases.py auxiva >> $out/call_phases.txt 2>/dev/null
python benchmarks/tools/call_phases.py fmnmf >> $out/call_phases.txt 2>/dev/null
tail -c 600 $out/bench.json
