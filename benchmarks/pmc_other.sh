#!/bin/bash
# SQ-level counter passes (rocprofv3 --pmc only, no tracing) for the kernels behind the bench line
# and behind configs[2] / configs[3] / GaussMNMF at 32 mixtures:
#   benchmarks/pmc_other.sh <tag>        (on the GPU box, through gpurun)
# Three passes per workload (at most 8 SQ counters fit one pass).  Writes gpurun_out/<tag>/<workload>_<pass>.json;
# `python benchmarks/sq_digest.py gpurun_out/<tag>/<workload>_*.json` merges them per kernel.
set -u
tag=${1:-r03_sq}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
PASS_a="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
PASS_b="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"
PASS_c="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
run() {  # <workload> <command...>
  local name=$1; shift
  for p in a b c; do
    local counters=PASS_$p
    rm -rf $out/${name}_$p
    timeout 300 rocprofv3 --pmc ${!counters} --output-format csv -d $out/${name}_$p -- "$@" > $out/${name}_$p.log 2>&1 \
      || tail -3 $out/${name}_$p.log
    python benchmarks/pmc_summary.py $out/${name}_$p > $out/${name}_$p.json
  done
}
run ilrma_b128 python bench.py --batch 128 --steps 2 --warmup 1 --no-cpu-baseline --no-single
run iva_iss_b32 python benchmarks/other_configs.py --batch 32 --only iva_iss --iters 2
run fastmnmf_b32 python benchmarks/other_configs.py --batch 32 --only fastmnmf --iters 2
run gmnmf_b32 python benchmarks/other_configs.py --batch 32 --only gmnmf --iters 4
ls $out/*.json
