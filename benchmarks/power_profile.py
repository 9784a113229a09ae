#!/usr/bin/env python3
"""Socket power and shader clock while each pass kernel of the bench line runs alone (GPU box).

    python benchmarks/power_profile.py [--seconds 2.0] [--batch 128]

For each of {basis, activation, covariance, whole iteration} the kernel is launched back to back for
`--seconds` while a host thread samples the hwmon power sensor and the current shader clock
(amdgpu sysfs; `rocm-smi` as a fall-back).  Prints one JSON object.  Round 4 finding: the ILRMA
passes run at the 1400 W package cap and the shader clock sags from 2400 to ~2050 MHz -- the bound
of the bench line is socket power, not HBM bandwidth and not instruction issue."""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _sysfs_cards():
    """[(power path, sclk path, cap path)] of every amdgpu card with a power sensor.  A box can expose
    more cards than the one HIP sees: the sampler reads all of them and reports the busiest."""
    cards = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            power = next((os.path.join(hw, n) for n in ("power1_average", "power1_input")
                          if os.path.exists(os.path.join(hw, n))), None)
            if not power:
                continue
            cap = os.path.join(hw, "power1_cap")
            f = os.path.join(hw, "freq1_input")
            cards.append((power, f if os.path.exists(f) else None,
                          cap if os.path.exists(cap) else None))
    return cards


def _read_number(path):
    try:
        return float(open(path).read().strip())
    except (OSError, ValueError):
        return None


class Sampler:
    """Samples (power W, sclk MHz) of every card every `period` seconds until stopped."""

    def __init__(self, period=0.05):
        self.cards = _sysfs_cards()
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        self.source = "amdgpu hwmon sysfs ({} cards)".format(len(self.cards))

    def _run(self):
        while not self._stop.is_set():
            self.samples.append([(_read_number(p), _read_number(f) if f else None)
                                 for p, f, _ in self.cards])
            time.sleep(self.period)

    def __enter__(self):
        self.samples = []
        self._stop.clear()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()

    def busiest(self):
        means = [np.mean([s[i][0] or 0.0 for s in self.samples]) for i in range(len(self.cards))]
        return int(np.argmax(means)) if means else None

    def summary(self, skip=0.25, card=None):
        """Mean over the samples after the first `skip` fraction (the ramp), of `card` (default: the
        card that drew the most power over the leg)."""
        if not self.cards or not self.samples:
            return {"socket_w": None, "sclk_mhz": None, "samples": 0, "source": self.source}
        i = self.busiest() if card is None else card
        s = self.samples[int(len(self.samples) * skip):]
        pw = [t[i][0] for t in s if t[i][0]]
        ck = [t[i][1] for t in s if t[i][1]]
        cap = _read_number(self.cards[i][2]) if self.cards[i][2] else None
        return {"socket_w": round(float(np.mean(pw)) / 1e6, 1) if pw else None,
                "socket_w_max": round(float(np.max(pw)) / 1e6, 1) if pw else None,
                "sclk_mhz": round(float(np.mean(ck)) / 1e6, 0) if ck else None,
                "cap_w": cap / 1e6 if cap else None, "card": i, "samples": len(s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--no-others", action="store_true", help="skip the configs[2] / configs[3] legs")
    args = ap.parse_args()
    import bench
    from ssspy_amd import _device as dv
    from ssspy_amd import _ops
    from ssspy_amd.utils.dataset import nmf_mixture_batch

    torch.cuda.set_device(0)
    N, F, T, K, B = 4, 1025, 512, 16, args.batch
    X = torch.from_numpy(nmf_mixture_batch(1000, B, N, F, T)).to("cuda:0")
    sep = bench.make_separator(X, K, seed=2000)
    for _ in range(3):
        sep.update_once()
    sep._U = dv.empty((B, F, N, N, N), dv.c128, X.device)

    def wcov():
        _ops.ilrma_weighted_covariance(sep._X, sep._state_dev("basis"), sep._state_dev("activation"),
                                       float(sep.domain), sep._ws, sep._ws_bytes, out=sep._U)

    Xr = torch.view_as_real(X)
    acc = torch.zeros((), dtype=torch.float64, device=X.device)

    def stream():  # a read-only pass over the same 4.3 GB with next to no arithmetic (torch reduction)
        acc.copy_(Xr.sum())

    legs = {"iteration": sep.update_once, "basis": sep.update_basis_mm,
            "activation": sep.update_activation_mm, "covariance": wcov, "read_stream": stream}
    sampler = Sampler()
    out = {"batch": B, "source": sampler.source}
    torch.cuda.synchronize()
    card = None
    for name, fn in legs.items():
        fn()
        torch.cuda.synchronize()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with sampler:
            t0 = time.perf_counter()
            e0.record()
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(20):
                    fn()
                n += 20
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        if card is None:
            card = sampler.busiest()  # the card HIP runs on: the one the first leg loads
        rec = sampler.summary(0.5, card)  # second half of the leg: the sensor has settled
        rec["ms_per_launch"] = round(ms, 4)
        passes = 3 if name == "iteration" else 1
        rec["hbm_TBs"] = round(passes * 16.0 * N * F * T * B / (ms * 1e-3) / 1e12, 3)
        out[name] = rec
    sep._check_device_errors()
    del sep, X, Xr
    torch.cuda.empty_cache()
    if not args.no_others:
        from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast
        from ssspy_amd.bss.mnmf import FastGaussMNMF

        def leg(name, m, bytes_per_step):
            for _ in range(3):
                m.update_once()
            torch.cuda.synchronize()
            n = 0
            with sampler:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < args.seconds:
                    for _ in range(10):
                        m.update_once()
                    n += 10
                    torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            rec = sampler.summary(0.5, card)
            rec["ms_per_step"] = round(1e3 * dt / n, 4)
            rec["hbm_TBs"] = round(bytes_per_step / (dt / n) / 1e12, 3)
            out[name] = rec
            m._check_device_errors()

        N2, F2, T2, B2 = 8, 2049, 1024, 32
        X2 = torch.from_numpy(nmf_mixture_batch(3000, B2, N2, F2, T2)).to("cuda:0")
        m = AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=False)
        m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
        m._bind_input(X2)
        m._reset()
        leg("configs2_iss_b32", m, 2 * 16.0 * N2 * F2 * T2 * B2)
        del m, X2
        torch.cuda.empty_cache()
        B3 = 128
        X3 = torch.from_numpy(nmf_mixture_batch(4000, B3, 4, 1025, 512)).to("cuda:0")
        m = FastGaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))
        m._bind_input(X3)
        m._reset()
        leg("configs3_fmnmf_b128", m, 4 * 16.0 * 4 * 1025 * 512 * B3)
        m = AuxLaplaceIVA(spatial_algorithm="IP", record_loss=False)
        m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
        m._bind_input(X3)
        m._reset()
        m._C()
        leg("auxiva_ip_b128", m, 2 * 16.0 * 4 * 1025 * 512 * B3)
        del m, X3
        torch.cuda.empty_cache()
    time.sleep(5.0)  # the hwmon power value is a moving average: let the load drain out of it
    with sampler:
        time.sleep(1.0)
    out["idle"] = sampler.summary(0.0, card)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
