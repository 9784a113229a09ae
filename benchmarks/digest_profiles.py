#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by benchmarks/profile_round.sh) into the tracked files under
profiles/: the bench line, the rocprofv3 per-kernel statistics, and the HBM traffic of the three
pass kernels per launch (FETCH_SIZE / WRITE_SIZE in KB; the read side is doubled on gfx950 as
MI355X_MICROARCH.md prescribes: FETCH_SIZE tallies 128-byte requests as 64 bytes).

    python benchmarks/digest_profiles.py <tag>
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")

bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
B = bench["config"]["batch_per_gpu"]
json.dump(bench, open(os.path.join(dst, "{}_bench_b{}.json".format(tag, B)), "w"), indent=1)



def latest(pattern):
    """gpurun merges every run's files into the same directory: take the newest match only."""
    found = glob.glob(pattern, recursive=True)
    return max(found, key=os.path.getmtime) if found else None


stats = latest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
if stats:
    rows = [r for r in csv.DictReader(open(stats))]
    keep = [r for r in rows if float(r.get("Percentage", 0) or 0) >= 0.01]
    with open(os.path.join(dst, "{}_bench_b{}_kernel_stats.csv".format(tag, B)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            r["Name"] = r["Name"][:120]
            w.writerow(r)

KERNELS = {"k_basis_fast": "k_ilrma_basis", "k_activation_fast": "k_ilrma_activation",
           "k_wcov_fast": "k_ilrma_wcov"}


def counter(path, name):
    acc = collections.defaultdict(list)
    newest = latest(os.path.join(src, path, "**", "*counter_collection.csv"))
    for p in [newest] if newest else []:
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != name:
                continue
            for k in KERNELS:
                if "::" + k in r["Kernel_Name"] or r["Kernel_Name"].startswith(k):
                    acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if v}


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
algo = 16.0 * bench["config"]["n_sources"] * bench["config"]["n_bins"] * bench["config"]["n_frames"] * B
traffic = {}
for k, label in KERNELS.items():
    if k in fetch and k in write:
        hbm = (2.0 * fetch[k] + write[k]) * 1024.0
        traffic[label] = {
            "kernel": k, "FETCH_SIZE_KB_raw": round(fetch[k]), "WRITE_SIZE_KB_raw": round(write[k]),
            "hbm_bytes_per_launch_batch{}".format(B): round(hbm),
            "algorithmic_bytes_per_launch_batch{}".format(B): round(algo),
            "ratio": round(hbm / algo, 3),
            "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no tracing), "
                    "bench.py --batch {}; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE "
                    "tallies 128-B requests as 64 B)".format(B),
        }
if traffic:
    traffic["_round"] = tag
    sys.path.insert(0, ROOT)
    from ssspy_amd.utils.dataset import kernel_sources_sha256
    traffic["_kernel_sources_sha256"] = kernel_sources_sha256()  # bench.py: stale after a kernel change
    json.dump(traffic, open(os.path.join(dst, "roofline_traffic.json"), "w"), indent=1)
other = latest(os.path.join(src, "other_stats", "**", "*kernel_stats.csv"))
if other:
    rows = [r for r in csv.DictReader(open(other))]
    with open(os.path.join(dst, "{}_other_configs_b32_kernel_stats.csv".format(tag)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.01:
                r["Name"] = r["Name"][:120]
                w.writerow(r)
oc = os.path.join(src, "other_configs.txt")
if os.path.exists(oc):
    shutil.copy(oc, os.path.join(dst, "{}_other_configs.txt".format(tag)))
single = os.path.join(src, "single.txt")
if os.path.exists(single):  # single-mixture timelines (benchmarks/single_trace.py) + the AuxIVA lines
    keep = [ln for ln in open(single) if "amdgpu.ids" not in ln]
    open(os.path.join(dst, "{}_single_mixture.txt".format(tag)), "w").writelines(keep)
# round 5 additions (benchmarks/profile_round.sh)
for name, out_name in (("mnmf_steps.txt", "{}_mnmf_steps.txt"), ("call_timeline.txt", "{}_call_timeline.txt"),
                       ("cache_energy.json", "{}_cache_energy.json"),
                       ("subbatch_sweep.txt", "{}_subbatch_sweep.txt"),
                       ("batch_sweep.txt", "{}_batch_sweep.txt"),
                       ("power_profile.json", "{}_power_profile.json")):
    path = os.path.join(src, name)
    if os.path.exists(path) and os.path.getsize(path):
        keep = [ln for ln in open(path) if "amdgpu.ids" not in ln]
        open(os.path.join(dst, out_name.format(tag)), "w").writelines(keep)
for path in sorted(glob.glob(os.path.join(src, "legs", "*_b32_kernel_stats.csv"))):
    rows = [r for r in csv.DictReader(open(path))]
    leg = os.path.basename(path)[: -len("_b32_kernel_stats.csv")]
    with open(os.path.join(dst, "{}_pairwise_ipa_{}_b32_kernel_stats.csv".format(tag, leg)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.05:
                r["Name"] = r["Name"][:120]
                w.writerow(r)
for path in sorted(glob.glob(os.path.join(src, "legs", "*_n8_b16_kernel_stats.csv"))):
    rows = [r for r in csv.DictReader(open(path))]
    leg = os.path.basename(path)[: -len("_n8_b16_kernel_stats.csv")]
    with open(os.path.join(dst, "{}_legs8_{}_b16_kernel_stats.csv".format(tag, leg)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.05:
                r["Name"] = r["Name"][:120]
                w.writerow(r)
path = os.path.join(src, "legs_ms.txt")
if os.path.exists(path) and os.path.getsize(path):
    open(os.path.join(dst, "{}_legs_ms.txt".format(tag)), "w").writelines(
        ln for ln in open(path) if "amdgpu.ids" not in ln)
call = latest(os.path.join(src, "call_stats", "**", "*kernel_stats.csv"))
if call:
    rows = [r for r in csv.DictReader(open(call))]
    with open(os.path.join(dst, "{}_call_kernel_stats.csv".format(tag)), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.05:
                r["Name"] = r["Name"][:120]
                w.writerow(r)
print(json.dumps({"bench_value": bench["value"], "roofline": bench["roofline"], "traffic": traffic}, indent=1))
# round 6 additions: FastGaussMNMF above 4 channels, the leg survey, GaussMNMF per channel count, the
# phases of one call of configs[1..3]
for name, out_name in (("fmnmf_wide.txt", "{}_fmnmf_wide.txt"), ("leg_survey.txt", "{}_leg_survey.txt"),
                       ("gmnmf_channels.txt", "{}_gmnmf_channels.txt"),
                       ("call_phases.txt", "{}_call_phases.txt")):
    path = os.path.join(src, name)
    if os.path.exists(path) and os.path.getsize(path):
        keep = [ln for ln in open(path) if "amdgpu.ids" not in ln]
        open(os.path.join(dst, out_name.format(tag)), "w").writelines(keep)
for path in sorted(glob.glob(os.path.join(src, "fmnmf_wide_m*_kernel_stats.csv"))):
    rows = [r for r in csv.DictReader(open(path))]
    with open(os.path.join(dst, "{}_{}".format(tag, os.path.basename(path))), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            if float(r.get("Percentage", 0) or 0) >= 0.05:
                r["Name"] = r["Name"][:120]
                w.writerow(r)
