#!/usr/bin/env python3
"""Classified instruction audit of the innermost loop of one kernel in a hipcc -S listing.

    hipcc <flags of _build.py> --cuda-device-only -S -o unit.s ssspy_amd/csrc/<unit>.hip
    python benchmarks/tools/isa_audit.py unit.s <mangled-name-substring> [--list]

Classes: what a tile costs on the SIMD's issue port (fp64 VALU at 4 clocks per wave instruction,
v_rcp_f64 / transcendental at 16, v_mfma_f64_16x16x4 at 64) and everything that is not arithmetic
of the algorithm (moves, selects, compares, integer / address, conversions).  --list prints the
non-arithmetic opcodes with their counts."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
hdrs = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
hdr = max(hdrs)
# main line of the loop: from the header to the first unconditional branch back to a label laid out
# at or before the header (the latch); blocks laid out behind it are the compiler's out-of-line
# alternatives (e.g. the n_basis < 16 k-slab skips), not executed on the audited configuration
labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
lo = hdr
hi = next(i for i in range(hdr + 1, len(body))
          if re.match(r"\s*s_branch\s+(\S+)", body[i])
          and labels.get(re.match(r"\s*s_branch\s+(\S+)", body[i]).group(1), len(body)) <= hdr)

def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_pk_fma_f64"):
        return "valu_f64_arith"
    if op in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64",
              "v_div_fixup_f64", "v_frexp_mant_f64", "v_frexp_exp_i32_f64", "v_ldexp_f64",
              "v_trig_preop_f64", "v_fract_f64", "v_rndne_f64", "v_floor_f64"):
        return "valu_f64_special"
    if op.startswith("v_max_f64") or op.startswith("v_min_f64"):
        return "valu_f64_minmax"
    if op.startswith("v_cndmask"):
        return "valu_select"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "valu_compare"
    if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readlane") \
            or op.startswith("v_readfirstlane") or op.startswith("v_writelane") \
            or op.startswith("v_permlane") or op.startswith("v_swap"):
        return "valu_move"
    if op.startswith("v_cvt"):
        return "valu_convert"
    if op.startswith("v_"):
        return "valu_int_addr"
    return "other"


mix = collections.Counter()
ops = collections.defaultdict(collections.Counter)
for l in body[lo:hi + 1]:
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith(".") or t[0].endswith(":"):
        continue
    op = t[0]
    base = re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)
    c = classify(base)
    mix[c] += 1
    ops[c][base] += 1

arith = mix["valu_f64_arith"] + mix["valu_f64_special"] + mix["valu_f64_minmax"]
non = sum(mix[k] for k in ("valu_select", "valu_compare", "valu_move", "valu_convert", "valu_int_addr"))
print("kernel {} loop lines {}..{}".format(key, lo, hi))
for k in sorted(mix):
    print("  {:18s} {:5d}".format(k, mix[k]))
print("  VALU total {}  (fp64 arithmetic {}, non-arithmetic {})".format(arith + non, arith, non))
clk = 4 * mix["valu_f64_arith"] + 4 * mix["valu_f64_minmax"] + 16 * mix["valu_f64_special"] \
    + 64 * mix["mfma"] + 4 * non
print("  issue clocks per trip (4 / fp64 VALU, 16 / special, 64 / MFMA, 4 / other VALU): {}".format(clk))
if "--list" in sys.argv:
    for k in ("valu_f64_arith", "valu_f64_special", "valu_select", "valu_compare", "valu_move",
              "valu_convert", "valu_int_addr", "lds", "vmem", "scratch", "salu"):
        if ops[k]:
            print("  [{}] {}".format(k, ", ".join("{} x{}".format(o, n) for o, n in ops[k].most_common())))
