#!/usr/bin/env python3
"""Merge the SQ-level counter passes of benchmarks/pmc_pass.sh into one per-kernel digest.

    python benchmarks/sq_digest.py gpurun_out/r03_pmc_a.json gpurun_out/r03_pmc_b.json > profiles/r03_pmc_sq_digest.json

Inputs: the JSON summaries benchmarks/pmc_summary.py writes per counter pass (per kernel: counter
sums averaged over the launches).  Output per kernel: the raw counters plus the ratios DESIGN.md
quotes -- shares of wave cycles (waiting for an instruction's operands / waiting at s_waitcnt /
issuing), instructions per wave, matrix-core cycles per wave.  SQ_WAVE_CYCLES and the SQ_WAIT_* /
SQ_ACTIVE_* counters tick once per four clocks (checked against the kernel durations: waves x
lifetime / resident wave slots = launch time); SQ_VALU_MFMA_BUSY_CYCLES is in clocks (= 64 per
v_mfma_f64_16x16x4, exactly).
"""
import json
import sys

merged = {}
for path in sys.argv[1:]:
    for kernel, counters in json.load(open(path)).items():
        merged.setdefault(kernel, {}).update(counters)

out = {}
for kernel, c in merged.items():
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
    out[kernel] = d
json.dump(out, sys.stdout, indent=1, sort_keys=True)
print()
