#!/usr/bin/env python3
"""Merge the SQ-level counter passes of benchmarks/pmc_pass.sh into one per-kernel digest.

    python benchmarks/sq_digest.py gpurun_out/r03_sq/ilrma_b128_?.json > one_workload.json
    python benchmarks/sq_digest.py --all gpurun_out/r03_sq > profiles/r03_pmc_sq_digest.json
(--all: every <workload>_<pass>.json that benchmarks/pmc_other.sh wrote, one top-level key per workload)

Inputs: the JSON summaries benchmarks/pmc_summary.py writes per counter pass (per kernel: counter
sums averaged over the launches).  Output per kernel: the raw counters plus the ratios DESIGN.md
quotes -- shares of wave cycles (waiting for an instruction's operands / waiting at s_waitcnt /
issuing), instructions per wave, matrix-core cycles per wave.  SQ_WAVE_CYCLES and the SQ_WAIT_* /
SQ_ACTIVE_* counters tick once per four clocks (checked against the kernel durations: waves x
lifetime / resident wave slots = launch time); SQ_VALU_MFMA_BUSY_CYCLES is in clocks (= 64 per
v_mfma_f64_16x16x4, exactly).
"""
import glob
import json
import os
import re
import sys

def digest(paths):
    merged = {}
    for path in paths:
        for kernel, counters in json.load(open(path)).items():
            merged.setdefault(kernel, {}).update(counters)
    return {kernel: one(c) for kernel, c in merged.items()}


def one(c):
    waves = c.get("SQ_WAVES", 0.0)
    cycles = c.get("SQ_WAVE_CYCLES", 0.0)
    d = {"counters": c}
    if waves and cycles:
        d["per_wave"] = {
            "lifetime_clocks": round(4.0 * cycles / waves, 1),
            "mfma_pipe_clocks": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / waves, 1),
            "valu": round(c.get("SQ_INSTS_VALU", 0.0) / waves, 1),
            "mfma": round(c.get("SQ_INSTS_MFMA", 0.0) / waves, 1),
            "lds": round(c.get("SQ_INSTS_LDS", 0.0) / waves, 1),
            "salu": round(c.get("SQ_INSTS_SALU", 0.0) / waves, 1),
            "vmem_rd": round(c.get("SQ_INSTS_VMEM_RD", 0.0) / waves, 1),
        }
        d["share_of_wave_cycles"] = {
            "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0.0) / cycles, 3),
            "wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / cycles, 3),
            "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / cycles, 3),
            "wait_inst_lds": round(c.get("SQ_WAIT_INST_LDS", 0.0) / cycles, 4),
        }
    if "SQ_INSTS_VALU_FMA_F64" in c:
        # fp64 flop per launch: VALU adds / muls / FMAs over 64 lanes, one MFMA "MOP" = 512 flop
        # (v_mfma_f64_16x16x4 = 4 MOPs = 1024 FMAs); v_rcp / v_rsq / v_sqrt are not counted
        flop = 64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0)
                       + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]) \
            + 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        d["fp64_gflop_per_launch"] = round(flop / 1e9, 3)
    return d


if sys.argv[1:2] == ["--all"]:
    root = sys.argv[2]
    names = sorted({re.sub(r"_[a-z]\.json$", "", os.path.basename(p))
                    for p in glob.glob(os.path.join(root, "*_[a-z].json"))})
    out = {w: digest(sorted(glob.glob(os.path.join(root, w + "_[a-z].json")))) for w in names}
else:
    out = digest(sys.argv[1:])
json.dump(out, sys.stdout, indent=1, sort_keys=True)
print()
