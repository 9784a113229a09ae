"""Host-side logic that needs no device: flooring translation, pair selectors, argument checks."""

import functools

import numpy as np
import pytest

from ssspy_amd import _lib
from ssspy_amd.bss.base import IterativeMethodBase
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.bss.iva import AuxGaussIVA, AuxIVA, AuxLaplaceIVA
from ssspy_amd.special.flooring import add_flooring, identity, max_flooring
from ssspy_amd.utils.flooring import choose_flooring_fn, device_flooring
from ssspy_amd.utils.select_pair import combination_pair_selector, sequential_pair_selector


def test_device_flooring_mapping():
    assert device_flooring(None) == (_lib.FLOOR_NONE, 0.0)
    assert device_flooring(identity) == (_lib.FLOOR_NONE, 0.0)
    assert device_flooring(max_flooring) == (_lib.FLOOR_MAX, 1e-10)
    assert device_flooring(functools.partial(max_flooring, eps=1e-5)) == (_lib.FLOOR_MAX, 1e-5)
    assert device_flooring(functools.partial(add_flooring, eps=1e-3)) == (_lib.FLOOR_ADD, 1e-3)
    # any other callable is kept for the host (evaluated on the small arrays it floors)
    fn = lambda x: x + 1  # noqa: E731
    floor = device_flooring(fn, allow_host=True)
    assert floor == (_lib.FLOOR_NONE, 0.0) and floor.host is fn
    assert device_flooring(max_flooring).host is None
    # ... only where the caller has a host-evaluation path; elsewhere it fails loudly instead of
    # running unfloored (round-3 advisor finding)
    with pytest.raises(NotImplementedError, match="to_psd"):
        device_flooring(fn, what="to_psd")
    # a separator keeps its resolved floor: deepcopy / pickle must keep the callable
    import copy
    import pickle

    assert copy.deepcopy(floor).host is fn
    kept = pickle.loads(pickle.dumps(device_flooring(abs, allow_host=True)))
    assert kept == (_lib.FLOOR_NONE, 0.0) and kept.host is abs


def test_choose_flooring_fn():
    class M:
        flooring_fn = staticmethod(max_flooring)

    assert choose_flooring_fn("self", method=M()) is not None
    assert choose_flooring_fn("self", method=None) is identity
    assert choose_flooring_fn(None) is identity


@pytest.mark.parametrize("n", [2, 3, 5])
def test_pair_selectors(n):
    assert list(sequential_pair_selector(n)) == [(m, (m + 1) % n) for m in range(n)]
    assert list(sequential_pair_selector(n, sort=True))[-1] == (0, n - 1)
    combos = list(combination_pair_selector(n))
    assert len(combos) == n * (n - 1) // 2 and all(a < b for a, b in combos)
    assert list(sequential_pair_selector(n, stop=2 * n, step=2))[0] == (0, 1)


def test_separators_are_iterative_methods():
    for m in (GaussILRMA(n_basis=2), AuxLaplaceIVA(), AuxGaussIVA(), AuxLaplaceIVA("ISS")):
        assert isinstance(m, IterativeMethodBase)
        assert m.loss == []
    assert GaussILRMA(n_basis=2, record_loss=False).loss is None


def test_constructor_argument_checks():
    with pytest.raises(AssertionError):
        GaussILRMA(n_basis=2, spatial_algorithm="XYZ")
    with pytest.raises(AssertionError):
        GaussILRMA(n_basis=2, domain=3)
    with pytest.raises(AssertionError):
        GaussILRMA(n_basis=2, newton_iter=3)  # IPA keyword without IPA
    ipa = GaussILRMA(n_basis=2, spatial_algorithm="IPA", newton_iter=3)
    assert ipa.newton_iter == 3 and ipa.lqpqm_normalization is True  # reference defaults
    with pytest.raises(AssertionError):
        GaussILRMA(n_basis=2, spatial_algorithm="IPA", bogus=1)
    assert AuxLaplaceIVA(spatial_algorithm="IPA").newton_iter == 1
    GaussILRMA(n_basis=2, flooring_fn=lambda x: x)  # any callable is accepted, as in the reference
    with pytest.raises(ValueError):
        GaussILRMA(n_basis=2, reference_id=None)
    with pytest.raises(AssertionError):
        AuxLaplaceIVA(spatial_algorithm="nope")


def test_generic_auxiva_contrast_resolution():
    """Tagged built-ins run in the kernels; user closures are host-evaluated on the frame norms
    (resolved to None here); a missing d_contrast_fn is an argument error before any device work."""
    from ssspy_amd.bss.iva import _device_contrast

    lap = AuxLaplaceIVA()
    assert _device_contrast(lap.contrast_fn, lap.d_contrast_fn) == _lib.CONTRAST_LAPLACE
    assert _device_contrast(lambda y: y, lambda r: r) is None
    m = AuxIVA(contrast_fn=lambda y: y, d_contrast_fn=None)
    with pytest.raises(ValueError):
        m(np.zeros((2, 4, 8), dtype=complex), n_iter=1)


def test_nmf_mixture_bytes_are_pinned():
    """The benchmark inputs are regenerated on every box from the seed; their SHA-256 is committed
     and the generator avoids BLAS / pow so the bytes do not depend on the CPU."""
    import json
    import os

    from ssspy_amd.utils.dataset import nmf_mixture, nmf_mixture_batch, sha256_of

    here = os.path.dirname(os.path.abspath(__file__))
    pins = json.load(open(os.path.join(here, "golden", "input_sha256.json")))
    for name in ("configs0_seed0_N2_F257_T128", "configs1_seed1000_N4_F1025_T512",
                 "configs3_seed4000_N4_F1025_T512"):
        pin = pins[name]
        assert sha256_of(nmf_mixture(pin["seed"], *pin["shape"])) == pin["sha256"], name
    batch = nmf_mixture_batch(0, 3, 2, 257, 128, workers=3)
    assert sha256_of(batch[0]) == pins["configs0_seed0_N2_F257_T128"]["sha256"]
    assert np.array_equal(batch[2], nmf_mixture(2, 2, 257, 128))


def test_run_sharded_skips_empty_blocks():
    from ssspy_amd.parallel import run_sharded, shard_bounds

    assert shard_bounds(1, 1, 2) == (1, 1)
    seen = []
    out = run_sharded(lambda lo, hi: seen.append((lo, hi)) or np.arange(lo, hi), 3)
    assert seen == [(0, 3)] and list(out) == [0, 1, 2]
    assert run_sharded(lambda lo, hi: 1 / 0, 0) is None  # nothing to do: never called


def test_callbacks_and_loss_protocol():
    calls = []

    class Dummy(IterativeMethodBase):
        def update_once(self):
            calls.append("u")

        def compute_loss(self):
            return float(len(calls))

    d = Dummy(callbacks=lambda m: calls.append("c"))
    d(n_iter=2)
    assert calls == ["c", "u", "c", "u", "c"]
    assert len(d.loss) == 3
    d2 = Dummy(record_loss=False)
    d2(n_iter=1, initial_call=False)
    assert d2.loss is None


def test_bench_gpus_flag_is_read():
    """bench.py --gpus N under a launcher whose WORLD_SIZE disagrees must fail before touching a
    device (the check sits in front of the HIP-device assertion, so it runs here too)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr


def test_named_windows_equal_scipy():
    """ssspy_amd.transform.get_window restates SciPy's periodic windows (a host-side table of n_fft
    samples); the reference's workflow passes the name straight to scipy.signal.stft."""
    import scipy.signal as ss

    from ssspy_amd.transform import get_window

    names = ["hann", "hamming", "blackman", "blackmanharris", "nuttall", "flattop", "boxcar",
             "bartlett", "triang", "cosine", "bohman", "parzen", ("kaiser", 8.6), ("gaussian", 7.0),
             ("tukey", 0.3), ("general_hamming", 0.6)]
    for n in (2, 9, 64, 441, 1000):
        for w in names:
            assert np.allclose(get_window(w, n), ss.get_window(w, n), rtol=0, atol=1e-13), (n, w)
    with pytest.raises(ValueError):
        get_window("nope", 16)


def test_pipeline_blocks_cover_the_batch_in_order():
    """parallel.pipeline_blocks: contiguous, disjoint, complete; no block above sub_batch; short
    edge blocks only when there are more than two sub-batches of work."""
    from ssspy_amd.parallel import pipeline_blocks

    for B in (1, 2, 5, 63, 64, 65, 128, 129, 256, 1000):
        for sub in (1, 3, 16, 64, 300):
            for ramp in (False, True):
                blocks = pipeline_blocks(B, sub, ramp)
                assert blocks[0][0] == 0 and blocks[-1][1] == B
                assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
                eff = max(1, min(sub, B))
                assert all(0 < hi - lo <= eff for lo, hi in blocks)
                if ramp and B > 2 * eff:
                    edge = max(1, eff // 4)
                    assert blocks[0] == (0, edge) and blocks[-1] == (B - edge, B)
                else:
                    assert all(hi - lo == eff for lo, hi in blocks[:-1])
    assert pipeline_blocks(256, 64) == [(0, 16), (16, 80), (80, 144), (144, 208), (208, 240),
                                        (240, 256)]


def test_lds_dma_tile_index_maps():
    """Index maps of k_mnmf_binmajor_glds (mnmf_kernels.hip), replayed on the host: the producer lane
    of instruction (m, quad) writes 16 bytes at lane * 16 of its 1 KB block; the consumer lane (c, q)
    must find frame 4 q + r of channel m and bin c, every 16-lane group of ds_read_b128 must touch
    16 different bank slots, and four adjacent producer lanes must fetch one 64-byte run."""
    M, T16 = 4, 16
    tile = np.arange(M * 16 * T16).reshape(M, 16, T16)  # (channel, bin, frame) sample ids
    lds = np.full(M * 4 * 64, -1)
    for m in range(M):
        for quad in range(4):
            for lane in range(64):
                cl = lane >> 2
                rl = (lane & 3) ^ ((cl >> 2) & 3)
                lds[(4 * m + quad) * 64 + lane] = tile[m, cl, 4 * quad + rl]
            frames = [[4 * quad + ((l & 3) ^ (((l >> 2) >> 2) & 3)) for l in range(4 * g, 4 * g + 4)]
                      for g in range(16)]
            assert all(sorted(f) == list(range(4 * quad, 4 * quad + 4)) for f in frames)
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    assert sorted(sum(groups, [])) == list(range(64))
    for m in range(M):
        for r in range(4):
            slot16 = {}
            for lane in range(64):
                c, q = lane & 15, lane >> 4
                byte = q * 1024 + c * 64 + 16 * (r ^ ((c >> 2) & 3)) + m * 4096
                assert byte % 16 == 0
                assert lds[byte // 16] == tile[m, c, 4 * q + r]
                slot16[lane] = (byte // 16) % 16
            for g in groups:
                assert len({slot16[l] for l in g}) == 16
    # activation ring: instruction (source n, half h) = rows 8 h .. 8 h + 7 of 16 frames (8 bytes each);
    # GEMM1's A operand of lane (c, q), k-slab ks is V[4 ks + q][tile_pi(c)]
    V = np.arange(4 * 16 * 16).reshape(4, 16, 16)
    vl = np.full(4 * 16 * 16, -1)
    for n in range(4):
        for h in range(2):
            for lane in range(64):
                k, j = 8 * h + (lane >> 3), 2 * (lane & 7)
                base = (n * 2048 + h * 1024 + lane * 16) // 8
                vl[base], vl[base + 1] = V[n, k, j], V[n, k, j + 1]
    for n in range(4):
        for ks in range(4):
            for lane in range(64):
                c, q = lane & 15, lane >> 4
                pi = 4 * (c & 3) + (c >> 2)
                byte = (q * 16 + pi) * 8 + n * 2048 + ks * 512
                assert vl[byte // 8] == V[n, 4 * ks + q, pi]


def test_eight_lane_jacobi_tournament_schedule():
    """The round-robin order of herm_rows8.hpp replayed on the host.  Every round rotates the pairs
    of POSITIONS (0,1) (2,3) (4,5) (6,7) and then position m takes the content of position
    RR_SRC[m] (the constant is read out of the header): seven rounds must meet each of the 28 index
    pairs exactly once and bring every index back to its place -- the kernels rely on eigenvalue k
    and column k of the eigenvector matrix being at position k again after every sweep -- and a
    full sweep of the rotations that schedule prescribes must converge like the cyclic Jacobi method
    it replaces (a Hermitian 8 x 8 to 1e-14 in under 8 sweeps)."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "ssspy_amd", "csrc", "herm_rows8.hpp")).read()
    body = re.search(r"constexpr unsigned RR_SRC = (.*?);", text, re.S).group(1)
    word = 0
    for term in body.replace("\n", " ").split("|"):
        term = term.strip().strip("()")
        if "<<" in term:
            v, sh = term.split("<<")
            word |= int(v.strip().rstrip("u"), 0) << int(sh.strip())
        else:
            word |= int(term.rstrip("u"), 0)
    src = [(word >> (4 * m)) & 7 for m in range(8)]
    assert sorted(src) == list(range(8)) and src[0] == 0

    where = list(range(8))  # index held by each position
    met = set()
    for _ in range(7):
        for k in range(4):
            met.add(frozenset((where[2 * k], where[2 * k + 1])))
        where = [where[src[m]] for m in range(8)]
    assert len(met) == 28
    assert where == list(range(8))

    # the rotations themselves, as the kernel forms them (rows: J^H (A J), both triangles kept)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
    A = A + A.conj().T
    A0 = A.copy()
    ref = np.linalg.eigvalsh(A)
    W = np.eye(8, dtype=complex)
    perm = list(range(8))
    for sweep in range(8):
        off = np.linalg.norm(A - np.diag(np.diag(A)))
        if off <= 1e-15 * np.linalg.norm(np.diag(A)):
            break
        for _ in range(7):
            J = np.eye(8, dtype=complex)
            for k in range(4):
                p, q = 2 * k, 2 * k + 1
                apq, app, aqq = A[p, q], A[p, p].real, A[q, q].real
                mag = abs(apq)
                if mag * mag < 1e-300:
                    continue
                u = apq / mag
                tau = (aqq - app) / (2 * mag)
                t = np.copysign(1.0, tau) / (abs(tau) + np.hypot(1.0, tau))
                cs = 1 / np.hypot(1.0, t)
                su = t * cs * u
                J[p, p] = J[q, q] = cs
                J[p, q] = su
                J[q, p] = -np.conj(su)
            A = J.conj().T @ A @ J
            W = W @ J
            P = np.zeros((8, 8))
            for m in range(8):
                P[src[m], m] = 1.0  # new position m <- old position src[m]
            A = P.T @ A @ P
            W = W @ P
            perm = [perm[src[m]] for m in range(8)]
        assert perm == list(range(8))
    assert sweep < 8
    lam = np.diag(A).real
    np.testing.assert_allclose(np.sort(lam), ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(W @ np.diag(lam) @ W.conj().T, A0, atol=1e-12)  # A0 = W diag(lam) W^H
    np.testing.assert_allclose(W.conj().T @ W, np.eye(8), atol=1e-13)
