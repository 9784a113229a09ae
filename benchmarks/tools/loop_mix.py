#!/usr/bin/env python3
"""Instruction mix of the innermost loop of one kernel in a hipcc --save-temps .s file.

    python benchmarks/tools/loop_mix.py <file.s> <mangled-name-substring>

The loop is taken as the layout range from the last "Inner Loop Header" label of the kernel to the
last branch back to that label."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
hdr = max(i for i, l in enumerate(body) if "Inner Loop Header" in l)
label = body[hdr].split(":")[0]
back = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\b", l))
lo, hi = (hdr, back) if back > hdr else (back, hdr)
mix = collections.Counter()
for l in body[lo:hi + 1]:
    t = l.strip().split()[0] if l.strip() else ""
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    if t.startswith("v_mfma"):
        mix["mfma"] += 1
    elif t.startswith("scratch_"):
        mix["scratch"] += 1
    elif t.startswith("ds_"):
        mix["lds"] += 1
    elif t.startswith("global_") or t.startswith("buffer_"):
        mix["vmem"] += 1
    elif t.startswith("s_waitcnt"):
        mix["waitcnt"] += 1
    elif t.startswith("s_barrier"):
        mix["barrier"] += 1
    elif t.startswith("s_cbranch") or t.startswith("s_branch"):
        mix["branch"] += 1
    elif t.startswith("s_"):
        mix["salu"] += 1
    elif t.startswith("v_") and "f64" in t:
        mix["valu_f64"] += 1
    elif t.startswith("v_"):
        mix["valu_other"] += 1
print("lines {}..{} of the kernel".format(lo, hi), dict(mix))
