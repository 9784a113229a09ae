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
        out = subprocess.run(["c++filt"], input="\n".join(names),
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
    heavy = {}
    for r in rows:
        if int(r.get("VGPRs Spill", "0") or 0) > 64:
            heavy[r.get("unit", "?")] = heavy.get(r.get("unit", "?"), 0) + 1
    out.write("# hipcc -Rpass-analysis=kernel-resource-usage (gfx950); {} functions, {} spill more than 64 "
              "VGPRs{}\n".format(len(rows), sum(heavy.values()),
                                 (": " + ", ".join("{} {}".format(u, c) for u, c in sorted(heavy.items())))
                                 if heavy else ""))
    out.write("{:24s} {:64s} {:>5s} {:>5s} {:>6s} {:>7s} {:>5s} {:>4s} {:>7s}\n".format(
        "unit", "kernel", "VGPR", "AGPR", "spillV", "scratch", "SGPR", "occ", "LDS"))
    seen = set()
    for r, n in zip(rows, names):
        line = "{:64s} {:>5s} {:>5s} {:>6s} {:>7s} {:>5s} {:>4s} {:>7s}".format(
            short(n)[:64], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("TotalSGPRs", "?"),
            r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"))
        if line in seen:   # the helpers of common.hpp are compiled into every unit
            continue
        seen.add(line)
        out.write("{:24s} {}\n".format(r.get("unit", "")[:24], line))


def build_all():
    sys.path.insert(0, ROOT)
    from ssspy_amd import _build

    from concurrent.futures import ThreadPoolExecutor

    def one(unit):
        src, obj, extra = unit
        cmd = [_build._hipcc()] + _build.CXXFLAGS + extra + [
            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(_build.CSRC, src), "-o",
            "/tmp/_kr_" + obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            raise SystemExit(res.stderr[-2000:])
        open("/tmp/_kr_" + obj + ".log", "w").write(res.stderr)
        found = parse(res.stderr)
        for r in found:
            r["unit"] = obj.replace(".o", "")
        return found

    rows = []
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
        for found in pool.map(one, list(_build._units())):
            rows += found
    return rows


if __name__ == "__main__":
    if sys.argv[1:] == ["--build"]:
        table(build_all())
    else:
        rows = []
        for p in sys.argv[1:]:
            rows += parse(open(p).read())
        table(rows)
