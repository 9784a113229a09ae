#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip 2> log ; kernel_resources.py log [...]
or  kernel_resources.py --build      (compile every translation unit of the library, print the table)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names),
                             capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except Exception:
        return names


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: (?:.*?:)?\s*(Function Name|[A-Za-z /\[\]]+?):\s*(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


def short(name):
    name = re.sub(r"HIP_vector_type<double, 2u>", "c128", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("ssspy::", "")


def table(rows, out=sys.stdout):
    names = demangle([r["name"] for r in rows])
    out.write("{:70s} {:>5s} {:>5s} {:>6s} {:>5s} {:>4s} {:>7s}\n".format(
        "kernel", "VGPR", "AGPR", "spillV", "SGPR", "occ", "LDS"))
    for r, n in zip(rows, names):
        out.write("{:70s} {:>5s} {:>5s} {:>6s} {:>5s} {:>4s} {:>7s}\n".format(
            short(n)[:70], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
            r.get("SGPRs", "?"), r.get("Occupancy [waves/SIMD]", "?"),
            r.get("LDS Size [bytes/block]", "?")))


def build_all():
    sys.path.insert(0, ROOT)
    from ssspy_amd import _build

    rows = []
    for src, obj, extra in _build._units():
        cmd = [_build._hipcc()] + _build.CXXFLAGS + extra + [
            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(_build.CSRC, src), "-o",
            "/tmp/_kr_" + obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            raise SystemExit(res.stderr[-2000:])
        rows += parse(res.stderr)
    return rows


if __name__ == "__main__":
    if sys.argv[1:] == ["--build"]:
        table(build_all())
    else:
        rows = []
        for p in sys.argv[1:]:
            rows += parse(open(p).read())
        table(rows)
