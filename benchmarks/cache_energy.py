#!/usr/bin/env python3
"""Socket power and rate of a read-only stream whose working set sits in the 256 MB Infinity Cache
(64 / 128 / 192 MB, re-read) against the same stream out of HBM (4 GB): nJ per byte of each
(round-4 verdict item 6b -- under the 1400 W cap the headline's question is joules per byte, not
seconds per byte).  Needs benchmarks/micro/bin/cache_energy (built in the build container).

    python benchmarks/cache_energy.py [--seconds 4]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("power_profile",
                                               os.path.join(ROOT, "benchmarks", "power_profile.py"))
_pp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_pp)
Sampler = _pp.Sampler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    args = ap.parse_args()
    exe = os.path.join(ROOT, "benchmarks", "micro", "bin", "cache_energy")
    out = {"legs": []}
    # idle power first (device initialised, nothing running)
    import torch

    torch.cuda.init()
    torch.zeros(1, device="cuda")
    sampler = Sampler()
    with sampler:
        time.sleep(2.0)
    idle = sampler.summary(0.25)
    out["idle"] = idle
    card = idle.get("card")
    for mb in (64, 128, 192, 512, 4096):
        sampler = Sampler()
        with sampler:
            res = subprocess.run([exe, str(mb), str(args.seconds)], capture_output=True, text=True)
        rec = json.loads(res.stdout.strip().splitlines()[-1])
        pw = sampler.summary(0.5)
        rec.update({"socket_w": pw["socket_w"], "sclk_mhz": pw["sclk_mhz"], "card": pw["card"]})
        if pw["socket_w"] and rec["tb_per_s"]:
            rec["nj_per_byte"] = round(pw["socket_w"] / (rec["tb_per_s"] * 1e12) * 1e9, 4)
            if idle.get("socket_w"):
                rec["nj_per_byte_above_idle"] = round(
                    (pw["socket_w"] - idle["socket_w"]) / (rec["tb_per_s"] * 1e12) * 1e9, 4)
        out["legs"].append(rec)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
