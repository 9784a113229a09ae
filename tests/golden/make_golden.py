#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference (ssspy 0.2.0).

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference:

    python tests/golden/make_golden.py

It imports the reference, runs the hot-path separators and operators on small
seeded inputs and writes ``tests/golden/*.npz`` (inputs, explicit initial state,
state after iterations 1, 2 and 10 captured through the reference's own
callback hook, the full loss list and the final output).  Only the data is
committed; the reference's source never leaves /root/reference.
"""

import functools
import os
import sys

import numpy as np

REFERENCE = os.environ.get("SSSPY_REFERENCE", "/root/reference")
sys.path.insert(0, REFERENCE)

from ssspy.algorithm import minimal_distortion_principle, projection_back  # noqa: E402
from ssspy.bss._update_spatial_model import (  # noqa: E402
    update_by_ip1, update_by_ip2, update_by_ipa, update_by_iss1, update_by_iss2)
from ssspy.bss.ilrma import GGDILRMA, TILRMA, GaussILRMA  # noqa: E402
from ssspy.bss.iva import AuxGaussIVA, AuxIVA, AuxLaplaceIVA  # noqa: E402
from ssspy.bss.mnmf import FastGaussMNMF, GaussMNMF  # noqa: E402
from ssspy.linalg import eigh2, inv2  # noqa: E402
from ssspy.special.flooring import add_flooring, max_flooring  # noqa: E402
from ssspy.special.psd import to_psd  # noqa: E402
from ssspy.utils.select_pair import combination_pair_selector  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SNAP_ITERS = (1, 2, 10)


def gen_iid(seed, N, F, T):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))


def gen_mixture(seed, N, F, T, Kt=4):
    """Structured NMF-source mixture."""
    rng = np.random.default_rng(seed)
    R = (rng.random((N, F, Kt)) ** 4) @ (rng.random((N, Kt, T)) ** 4) + 1e-3
    g1 = rng.standard_normal((N, F, T))
    g2 = rng.standard_normal((N, F, T))
    S = np.sqrt(R / 2) * (g1 + 1j * g2)
    A = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
    return (A @ S.transpose(1, 0, 2)).transpose(1, 0, 2)


def flooring_of(spec):
    kind, eps = spec
    if kind == "max":
        return functools.partial(max_flooring, eps=eps)
    if kind == "add":
        return functools.partial(add_flooring, eps=eps)
    return None


class Snapshots:
    """Callback that copies the separator state after selected iterations."""

    def __init__(self, names):
        self.names = names
        self.count = -1  # the initial call happens before the first iteration
        self.store = {}

    def __call__(self, method):
        self.count += 1
        if self.count in SNAP_ITERS:
            for name in self.names:
                value = getattr(method, name, None)
                if value is not None:
                    self.store["it{}_{}".format(self.count, name)] = np.array(value, copy=True)


ONLY = sys.argv[1:]  # optional name prefixes: regenerate only those fixtures


def skipped(name):
    return bool(ONLY) and not any(name.startswith(p) for p in ONLY)


OUT_DIR = os.environ.get("SSSPY_GOLDEN_OUT", HERE)


def save(name, **arrays):
    """An .npz whose bytes depend on the arrays only (fixed member timestamps, sorted keys), so
    re-running this script reproduces the committed fixtures byte for byte."""
    import io
    import zipfile

    path = os.path.join(OUT_DIR, name + ".npz")
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
        for key in sorted(arrays):
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(arrays[key]), allow_pickle=False)
            info = zipfile.ZipInfo(key + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            zf.writestr(info, buf.getvalue())
    print("{:40s} {:8.1f} KiB".format(name, os.path.getsize(path) / 1024))


def meta(**kw):
    return {"meta_" + k: np.asarray(v) for k, v in kw.items()}


# --------------------------------------------------------------------------- ILRMA
def run_ilrma(name, *, N, F, T, K, algo, seed, gen=gen_iid, domain=2, flooring=("max", 1e-10),
              normalization=True, scale_restoration=True, n_iter=10, model=("gauss", None),
              source_algorithm="MM", partitioning=False, **ipa_kwargs):
    if skipped(name):
        return
    X = gen(seed, N, F, T)
    init = {}
    if partitioning:
        basis = np.random.default_rng(seed + 1).random((F, K))
        activation = np.random.default_rng(seed + 2).random((K, T))
        latent = np.random.default_rng(seed + 5).random((N, K))
        init["latent"] = latent / latent.sum(axis=0)
    else:
        basis = np.random.default_rng(seed + 1).random((N, F, K))
        activation = np.random.default_rng(seed + 2).random((N, K, T))
    snap = Snapshots(["demix_filter", "output", "basis", "activation", "latent"])
    common = dict(spatial_algorithm=algo, source_algorithm=source_algorithm, domain=domain,
                  partitioning=partitioning, flooring_fn=flooring_of(flooring),
                  callbacks=snap, normalization=normalization, scale_restoration=scale_restoration,
                  rng=np.random.default_rng(seed + 3))
    if model[0] == "t":
        m = TILRMA(n_basis=K, dof=model[1], **common)
    elif model[0] == "ggd":
        m = GGDILRMA(n_basis=K, beta=model[1], **common)
    else:
        m = GaussILRMA(n_basis=K, **common, **ipa_kwargs)
    Y = m(X, n_iter=n_iter, basis=basis, activation=activation,
          **{k: v.copy() for k, v in init.items()})
    out = dict(X=X, basis0=basis, activation0=activation, loss=np.array(m.loss), final_output=Y,
               final_basis=m.basis, final_activation=m.activation)
    if partitioning:
        out["latent0"] = init["latent"]
        out["final_latent"] = m.latent
    if m.demix_filter is not None:
        out["final_demix_filter"] = m.demix_filter
    out.update(snap.store)
    out.update(meta(kind="gauss_ilrma", algo=algo, domain=domain, n_basis=K, n_iter=n_iter,
                    model=model[0], model_param=(0.0 if model[1] is None else model[1]),
                    source_algorithm=source_algorithm, partitioning=partitioning,
                    floor_kind=flooring[0], floor_eps=flooring[1], normalization=normalization,
                    scale_restoration=scale_restoration,
                    newton_iter=ipa_kwargs.get("newton_iter", 1)))
    save(name, **out)


# --------------------------------------------------------------------------- IVA
def run_iva(name, *, N, F, T, algo, contrast, seed, gen=gen_iid, flooring=("max", 1e-10),
            scale_restoration=True, n_iter=10, keep_arrays=True):
    if skipped(name):
        return
    X = gen(seed, N, F, T)
    names = ["demix_filter", "output"] + (["variance"] if contrast == "gauss" else [])
    snap = Snapshots(names)
    cls = AuxLaplaceIVA if contrast == "laplace" else AuxGaussIVA
    m = cls(spatial_algorithm=algo, flooring_fn=flooring_of(flooring), callbacks=snap,
            scale_restoration=scale_restoration)
    Y = m(X, n_iter=n_iter)
    out = dict(loss=np.array(m.loss))
    if keep_arrays:
        out.update(X=X, final_output=Y)
        if m.demix_filter is not None:
            out["final_demix_filter"] = m.demix_filter
        out.update(snap.store)
    else:
        # known-answer scalars only (config-1 sized case): the input is regenerated from the seed
        out.update(kat_y000=Y[0, 0, 0], kat_energy=np.sum(np.abs(Y) ** 2),
                   kat_abs_sum=np.sum(np.abs(Y)))
    out.update(meta(kind="aux_iva", algo=algo, contrast=contrast, n_iter=n_iter, seed=seed,
                    shape=(N, F, T), floor_kind=flooring[0], floor_eps=flooring[1],
                    scale_restoration=scale_restoration))
    save(name, **out)


# --------------------------------------------------------------------------- MNMF
def run_mnmf(name, *, M, F, T, K, seed, n_sources=None, gen=gen_iid, flooring=("max", 1e-10),
             normalization=True, n_iter=10, diag_algo="IP", pairs="sequential"):
    if skipped(name):
        return
    N = M if n_sources is None else n_sources
    X = gen(seed, M, F, T)
    basis = np.random.default_rng(seed + 1).random((N, F, K))
    activation = np.random.default_rng(seed + 2).random((N, K, T))
    spatial = np.random.default_rng(seed + 4).random((F, N, M))
    snap = Snapshots(["diagonalizer", "spatial", "basis", "activation"])
    extra = {}
    if diag_algo != "IP":  # the IP1 fixtures keep their bytes: no new constructor arguments for them
        extra["diagonalizer_algorithm"] = diag_algo
        if pairs == "combination":
            extra["pair_selector"] = combination_pair_selector
    m = FastGaussMNMF(n_basis=K, n_sources=n_sources, flooring_fn=flooring_of(flooring),
                      callbacks=snap, normalization=normalization,
                      rng=np.random.default_rng(seed + 3), **extra)
    Y = m(X, n_iter=n_iter, basis=basis, activation=activation, spatial=spatial.copy())
    out = dict(X=X, basis0=basis, activation0=activation, spatial0=spatial,
               loss=np.array(m.loss), final_output=Y, final_basis=m.basis,
               final_activation=m.activation, final_diagonalizer=m.diagonalizer,
               final_spatial=m.spatial)
    out.update(snap.store)
    out.update(meta(kind="fast_gauss_mnmf", n_basis=K, n_sources=N, n_iter=n_iter,
                    floor_kind=flooring[0], floor_eps=flooring[1], normalization=normalization))
    if diag_algo != "IP":
        out.update(meta(diag_algo=diag_algo, pairs=pairs))
    save(name, **out)


def random_psd(seed, N, F, M):
    """Well-conditioned random Hermitian PSD matrices with unit trace, (N, F, M, M)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, F, M, M)) + 1j * rng.standard_normal((N, F, M, M))
    H = A @ A.swapaxes(-2, -1).conj() + 0.5 * np.eye(M)
    return H / np.real(np.trace(H, axis1=-2, axis2=-1))[..., None, None]


def run_gmnmf(name, *, M, F, T, K, seed, n_sources=None, gen=gen_iid, flooring=("max", 1e-10),
              normalization=True, n_iter=10, spatial_init=False, partitioning=False):
    if skipped(name):
        return
    N = M if n_sources is None else n_sources
    X = gen(seed, M, F, T)
    lead = () if partitioning else (N,)
    basis = np.random.default_rng(seed + 1).random(lead + (F, K))
    activation = np.random.default_rng(seed + 2).random(lead + (K, T))
    init = dict(basis=basis, activation=activation)
    if partitioning:
        latent = np.random.default_rng(seed + 5).random((N, K))
        init["latent"] = latent / latent.sum(axis=0)
    if spatial_init:
        init["spatial"] = random_psd(seed + 4, N, F, M)
    snap = Snapshots(["spatial", "basis", "activation", "latent"])
    m = GaussMNMF(n_basis=K, n_sources=n_sources, partitioning=partitioning,
                  flooring_fn=flooring_of(flooring),
                  callbacks=snap, normalization=normalization, rng=np.random.default_rng(seed + 3))
    Y = m(X, n_iter=n_iter, **{k: v.copy() for k, v in init.items()})
    out = dict(X=X, basis0=basis, activation0=activation, loss=np.array(m.loss), final_output=Y,
               final_basis=m.basis, final_activation=m.activation, final_spatial=m.spatial)
    if spatial_init:
        out["spatial0"] = init["spatial"]
    if partitioning:
        out["latent0"] = init["latent"]
        out["final_latent"] = m.latent
    out.update(snap.store)
    out.update(meta(kind="gauss_mnmf", n_basis=K, n_sources=N, n_iter=n_iter,
                    partitioning=partitioning,
                    floor_kind=flooring[0], floor_eps=flooring[1], normalization=normalization))
    save(name, **out)



# --------------------------------------------------------------------------- rng-drawn initial state
class InitialAndFinal:
    """Callback copying the state at the initial call (the rng-drawn parameters) and after the first
    and the last iteration."""

    def __init__(self, names, n_iter):
        self.names, self.n_iter = names, n_iter
        self.count = -1
        self.store = {}

    def __call__(self, method):
        self.count += 1
        if self.count in (0, 1, self.n_iter):
            for name in self.names:
                value = getattr(method, name, None)
                if value is not None:
                    self.store["it{}_{}".format(self.count, name)] = np.array(value, copy=True)


def run_rng_init(name, *, cls, seed, N, F, T, K, n_iter=5, gen=gen_iid, **kwargs):
    """No injected state: every parameter comes from ``rng=default_rng(seed + 3)`` in the order the
    reference draws it (ilrma.py:230-266: latent, basis, activation with partitioning; mnmf.py:
    221-254 basis, activation, latent; mnmf.py:535-538, 595: basis, activation, spatial)."""
    if skipped(name):
        return
    X = gen(seed, N, F, T)
    names = {"ilrma": ["latent", "basis", "activation", "demix_filter"],
             "fmnmf": ["basis", "activation", "diagonalizer", "spatial"],
             "gmnmf": ["basis", "activation", "latent", "spatial"]}[cls]
    snap = InitialAndFinal(names, n_iter)
    rng = np.random.default_rng(seed + 3)
    if cls == "ilrma":
        model = kwargs.pop("model", ("gauss", None))
        common = dict(n_basis=K, callbacks=snap, rng=rng, **kwargs)
        if model[0] == "t":
            m = TILRMA(dof=model[1], **common)
        else:
            m = GaussILRMA(**common)
        kwargs["model"], kwargs["model_param"] = model[0], (0.0 if model[1] is None else model[1])
    elif cls == "fmnmf":
        m = FastGaussMNMF(n_basis=K, callbacks=snap, rng=rng, **kwargs)
    else:
        m = GaussMNMF(n_basis=K, callbacks=snap, rng=rng, **kwargs)
    Y = m(X, n_iter=n_iter)
    out = dict(X=X, loss=np.array(m.loss), final_output=Y)
    out.update(snap.store)
    out.update(meta(kind="rng_init_" + cls, seed=seed, n_basis=K, n_iter=n_iter, **kwargs))
    save(name, **out)


def custom_floor(x):
    """A flooring callable that is none of the reference's three (ssspy/special/flooring.py)."""
    return np.maximum(x, 1e-8) + 1e-12


def run_custom_floor(name, *, kind, seed, N, F, T, K=4, n_iter=6, gen=gen_mixture, **kwargs):
    """``flooring_fn`` is an arbitrary callable in the reference (ilrma.py:70-89)."""
    if skipped(name):
        return
    X = gen(seed, N, F, T)
    rng = np.random.default_rng(seed + 3)
    if kind == "ilrma":
        snap = InitialAndFinal(["basis", "activation", "demix_filter", "output"], n_iter)
        m = GaussILRMA(n_basis=K, flooring_fn=custom_floor, callbacks=snap, rng=rng, **kwargs)
    elif kind == "tilrma":  # (round 6: the t model's weights hold no floor -- IP2 / ISS2 on the device)
        dof = kwargs.pop("dof")
        snap = InitialAndFinal(["basis", "activation", "demix_filter", "output"], n_iter)
        m = TILRMA(n_basis=K, dof=dof, flooring_fn=custom_floor, callbacks=snap, rng=rng, **kwargs)
        kwargs = dict(kwargs, model="t", model_param=dof)
        kind = "ilrma"
    elif kind == "ggdilrma":  # (round 6: GGD floors |y|^(2 - beta) per element of the spectrogram)
        beta = kwargs.pop("beta")
        snap = InitialAndFinal(["basis", "activation", "demix_filter", "output"], n_iter)
        m = GGDILRMA(n_basis=K, beta=beta, flooring_fn=custom_floor, callbacks=snap, rng=rng, **kwargs)
        kwargs = dict(kwargs, model="ggd", model_param=beta)
        kind = "ilrma"
    elif kind == "iva":
        snap = InitialAndFinal(["demix_filter", "output"], n_iter)
        m = AuxLaplaceIVA(flooring_fn=custom_floor, callbacks=snap, **kwargs)
    elif kind == "gaussiva":
        snap = InitialAndFinal(["demix_filter", "output", "variance"], n_iter)
        m = AuxGaussIVA(flooring_fn=custom_floor, callbacks=snap, **kwargs)
    else:
        snap = InitialAndFinal(["basis", "activation", "diagonalizer", "spatial"], n_iter)
        m = FastGaussMNMF(n_basis=K, flooring_fn=custom_floor, callbacks=snap, rng=rng, **kwargs)
    Y = m(X, n_iter=n_iter)
    out = dict(X=X, loss=np.array(m.loss), final_output=Y)
    out.update(snap.store)
    out.update(meta(kind="custom_floor_" + kind, seed=seed, n_basis=K, n_iter=n_iter, **kwargs))
    save(name, **out)


def run_eight_lane_operators():
    """Round 5: the reference on the sizes the device build runs with a matrix / bin on 8 lanes
    (herm_rows8.hpp): ssspy.linalg operators at 7 x 7 and 8 x 8, update_by_ipa at 7 and 8 sources."""
    if skipped("eight_lane_operators"):
        return
    from ssspy.linalg import eigh, gmeanmh, invsqrtmh, sqrtmh

    out = {}
    for M in (7, 8):
        rng = np.random.default_rng(300 + M)
        n = 20
        x = rng.standard_normal((n, M, 2 * M)) + 1j * rng.standard_normal((n, M, 2 * M))
        y = rng.standard_normal((n, M, 2 * M)) + 1j * rng.standard_normal((n, M, 2 * M))
        A = x @ x.swapaxes(-2, -1).conj() / (2 * M)
        B = y @ y.swapaxes(-2, -1).conj() / (2 * M)
        H = rng.standard_normal((n, M, M)) + 1j * rng.standard_normal((n, M, M))
        H = H + H.swapaxes(-2, -1).conj()  # indefinite
        k = "m{}_".format(M)
        out[k + "A"], out[k + "B"], out[k + "H"] = A, B, H
        out[k + "sqrtmh"] = sqrtmh(A)
        out[k + "invsqrtmh"] = invsqrtmh(A)
        out[k + "invsqrtmh_floor"] = invsqrtmh(A, flooring_fn=functools.partial(max_flooring, eps=0.6))
        out[k + "to_psd"] = to_psd(H)
        out[k + "to_psd_floor"] = to_psd(H, flooring_fn=functools.partial(max_flooring, eps=0.5))
        out[k + "to_psd_add"] = to_psd(H, flooring_fn=functools.partial(add_flooring, eps=0.25))
        for t in (1, 2, 3):
            out[k + "gmeanmh{}".format(t)] = gmeanmh(A, B, type=t)
            lamb, z = eigh(A, B, type=t)
            out[k + "eigh{}_lamb".format(t)] = lamb
            out[k + "eigh{}_z".format(t)] = z  # (phase of each column arbitrary)
    for N in (7, 8):
        rng = np.random.default_rng(320 + N)
        F, T = 33, 24  # (33 bins: a full block of 32 of the 8-lane kernel and a ragged one)
        Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        varphi = 1 / (rng.random((N, F, T)) + 0.1)
        k = "ipa{}_".format(N)
        out[k + "Y"], out[k + "varphi"] = Y, varphi
        out[k + "out"] = update_by_ipa(Y.copy(), varphi)
        out[k + "out_nonorm_it3"] = update_by_ipa(Y.copy(), varphi, normalization=False, max_iter=3)
        out[k + "out_bcast_add"] = update_by_ipa(
            Y.copy(), varphi[:, :1, :], flooring_fn=functools.partial(add_flooring, eps=1e-4))
        out[k + "out_it12"] = update_by_ipa(Y.copy(), varphi, max_iter=12)
    save("eight_lane_operators", **out)


def run_ipa_operators():
    """update_by_ipa on random spectrograms: per-source-count cases incl. broadcast weights."""
    if skipped("ipa_operators"):
        return
    out = {}
    for N in (2, 3, 4, 5):
        rng = np.random.default_rng(140 + N)
        F, T = 8, 36
        Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        varphi = 1 / (rng.random((N, F, T)) + 0.1)
        out["n{}_Y".format(N)] = Y
        out["n{}_varphi".format(N)] = varphi
        out["n{}_out".format(N)] = update_by_ipa(Y.copy(), varphi)
        out["n{}_out_nonorm_it3".format(N)] = update_by_ipa(Y.copy(), varphi, normalization=False,
                                                            max_iter=3)
        out["n{}_out_bcast_add".format(N)] = update_by_ipa(
            Y.copy(), varphi[:, :1, :], flooring_fn=functools.partial(add_flooring, eps=1e-4))
        # enough Newton steps for the loop to stop early (all bins converged): lqpqm.py:196-213
        out["n{}_out_it12".format(N)] = update_by_ipa(Y.copy(), varphi, max_iter=12)
    # the LQPQM solver itself, 10 Newton steps at most (its default)
    from ssspy.linalg import lqpqm2

    rng = np.random.default_rng(149)
    for L in (1, 2, 3, 5):
        n = 21
        A = rng.standard_normal((n, L, 9)) + 1j * rng.standard_normal((n, L, 9))
        H = A @ A.swapaxes(-2, -1).conj()
        H = H / np.real(np.trace(H, axis1=-2, axis2=-1))[:, None, None]
        v = rng.standard_normal((n, L)) + 1j * rng.standard_normal((n, L))
        z = rng.random(n) * 2.0
        out["lq{}_H".format(L)], out["lq{}_v".format(L)], out["lq{}_z".format(L)] = H, v, z
        out["lq{}_y".format(L)] = lqpqm2(H, v, z)
        out["lq{}_y_it2".format(L)] = lqpqm2(H, v, z, max_iter=2)
    save("ipa_operators", **out)


# --------------------------------------------------------------------------- boundary cases
POWER_CONTRAST_P = 1.5  # G_R(r) = r^p: neither of the two contrasts the kernels implement


def power_contrast_fn(y):
    return np.linalg.norm(y, axis=1) ** POWER_CONTRAST_P


def power_d_contrast_fn(r):
    return POWER_CONTRAST_P * r ** (POWER_CONTRAST_P - 1)


def run_generic_auxiva(name, *, N, F, T, algo, seed, gen=gen_iid, n_iter=10):
    """The generic AuxIVA class with user closures (ssspy/bss/iva.py:1582-1635)."""
    if skipped(name):
        return
    X = gen(seed, N, F, T)
    snap = Snapshots(["demix_filter", "output"])
    m = AuxIVA(spatial_algorithm=algo, contrast_fn=power_contrast_fn,
               d_contrast_fn=power_d_contrast_fn, callbacks=snap)
    Y = m(X, n_iter=n_iter)
    out = dict(X=X, loss=np.array(m.loss), final_output=Y)
    if m.demix_filter is not None:
        out["final_demix_filter"] = m.demix_filter
    out.update(snap.store)
    out.update(meta(kind="aux_iva_generic", algo=algo, n_iter=n_iter, power=POWER_CONTRAST_P))
    save(name, **out)


def run_all_channel_restoration():
    """projection_back / minimal_distortion_principle with reference_id=None (every channel)."""
    if skipped("restoration_all_channels"):
        return
    out = {}
    for N in (2, 3, 4):
        rng = np.random.default_rng(160 + N)
        F, T = 7, 26
        X = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        W = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
        out["n{}_X".format(N)], out["n{}_Y".format(N)], out["n{}_W".format(N)] = X, Y, W
        out["n{}_pb_filter".format(N)] = projection_back(W, reference_id=None)
        out["n{}_pb_output".format(N)] = projection_back(Y, reference=X, reference_id=None)
        out["n{}_mdp_output".format(N)] = minimal_distortion_principle(Y, reference=X,
                                                                      reference_id=None)
    save("restoration_all_channels", **out)


# --------------------------------------------------------------------------- operators
def run_operators():
    if skipped("operators"):
        return
    out = {}
    # update_by_ip1: seeds/shapes in the style of the reference's operator unit tests
    for N in (2, 3, 4, 8):
        rng = np.random.default_rng(40 + N)
        F, T = 9, 30
        X = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        W = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
        varphi = 1 / (rng.random((N, F, T)) + 0.1)
        XX = (X[:, None] * X[None].conj()).transpose(2, 0, 1, 3)
        U = np.mean(varphi.transpose(1, 0, 2)[:, :, None, None, :] * XX[:, None], axis=-1)
        out["ip1_n{}_W".format(N)] = W
        out["ip1_n{}_U".format(N)] = U
        out["ip1_n{}_out".format(N)] = update_by_ip1(W.copy(), U)
        out["ip1_n{}_out_add".format(N)] = update_by_ip1(
            W.copy(), U, flooring_fn=functools.partial(add_flooring, eps=1e-3))
        Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        out["iss1_n{}_Y".format(N)] = Y
        out["iss1_n{}_varphi".format(N)] = varphi
        out["iss1_n{}_out".format(N)] = update_by_iss1(Y, varphi)
        wt = varphi[:, :1, :]
        out["iss1_n{}_out_bcast".format(N)] = update_by_iss1(Y, wt)
        out["pb_n{}_filter".format(N)] = projection_back(W, reference_id=1)
        out["pb_n{}_output".format(N)] = projection_back(Y, reference=X, reference_id=0)
        out["pb_n{}_X".format(N)] = X
        A = rng.standard_normal((F, T, N, N)) + 1j * rng.standard_normal((F, T, N, N))
        H = A @ A.swapaxes(-2, -1).conj() - 0.5 * np.eye(N)  # indefinite Hermitian
        out["psd_n{}_in".format(N)] = H
        out["psd_n{}_out".format(N)] = to_psd(H)
    rng = np.random.default_rng(7)
    A = rng.standard_normal((16, 2, 2)) + 1j * rng.standard_normal((16, 2, 2))
    B = rng.standard_normal((16, 2, 2)) + 1j * rng.standard_normal((16, 2, 2))
    A = A @ A.swapaxes(-2, -1).conj()
    B = B @ B.swapaxes(-2, -1).conj()
    out["inv2_in"] = A
    out["inv2_out"] = inv2(A)
    lamb, z = eigh2(A, B)
    out["eigh2_A"], out["eigh2_B"] = A, B
    out["eigh2_lamb"], out["eigh2_z"] = lamb, z
    save("operators", **out)


def singular_below_half(x):
    """A singular_fn that is neither of the reference's two built-ins (ssspy/linalg/lqpqm.py:61-78)."""
    return x < 0.5


def run_lqpqm_singular():
    """lqpqm2 with singular_fn=None (||v|| == 0), a callable, and the default, on problems some of
    which have v exactly zero / tiny / small (ssspy/linalg/lqpqm.py:61-110)."""
    if skipped("lqpqm_singular"):
        return
    from ssspy.linalg import lqpqm2

    out = {}
    rng = np.random.default_rng(180)
    for L in (1, 2, 3, 5):
        n = 24
        A = rng.standard_normal((n, L, 7)) + 1j * rng.standard_normal((n, L, 7))
        H = A @ A.swapaxes(-2, -1).conj()
        H = H / np.real(np.trace(H, axis1=-2, axis2=-1))[:, None, None]
        v = rng.standard_normal((n, L)) + 1j * rng.standard_normal((n, L))
        v[::4] = 0.0                      # exactly zero: singular for every rule
        v[1::4] *= 1e-13                  # below the default floor (1e-10), not zero
        v[2::4] *= 0.2 / np.linalg.norm(v[2::4], axis=-1, keepdims=True)  # norm 0.2 < 0.5
        z = rng.random(n) * 2.0
        p = "l{}_".format(L)
        out[p + "H"], out[p + "v"], out[p + "z"] = H, v, z
        out[p + "y_default"] = lqpqm2(H, v, z)
        out[p + "y_none"] = lqpqm2(H, v, z, singular_fn=None)
        out[p + "y_callable"] = lqpqm2(H, v, z, singular_fn=singular_below_half)
        out[p + "y_none_nofloor_it3"] = lqpqm2(H, v, z, flooring_fn=None, singular_fn=None,
                                               max_iter=3)
    save("lqpqm_singular", **out)


def run_psd_custom_floor():
    """to_psd and invsqrtmh with a flooring callable that is none of the reference's three
    (ssspy/special/psd.py:11-71, ssspy/linalg/sqrtm.py:27-64)."""
    if skipped("psd_custom_floor"):
        return
    from ssspy.linalg import invsqrtmh

    out = {}
    for M in (2, 3, 4, 6):
        rng = np.random.default_rng(190 + M)
        A = rng.standard_normal((5, 7, M, M)) + 1j * rng.standard_normal((5, 7, M, M))
        H = A @ A.swapaxes(-2, -1).conj() - 0.5 * np.eye(M)      # indefinite Hermitian
        Hp = A @ A.swapaxes(-2, -1).conj() + 1e-9 * np.eye(M)    # positive definite, some tiny eigenvalues
        Hp[0, 0] = 1e-20 * np.eye(M)
        out["m{}_H".format(M)], out["m{}_Hp".format(M)] = H, Hp
        out["m{}_psd".format(M)] = to_psd(H, flooring_fn=custom_floor)
        out["m{}_invsqrt".format(M)] = invsqrtmh(Hp, flooring_fn=custom_floor)
    save("psd_custom_floor", **out)


def run_pairwise_operators():
    """update_by_ip2 / update_by_iss2 on random operands (ssspy/bss/_update_spatial_model.py:81-143,
    197-314): default (sequential) pairs, every combination, an explicit list with negative and
    descending indices (:241-251), broadcast weights, add-flooring."""
    if skipped("pairwise_operators"):
        return
    out = {}
    add = functools.partial(add_flooring, eps=1e-3)
    for N in (2, 3, 4, 5, 8):
        rng = np.random.default_rng(170 + N)
        F, T = 9, 34
        X = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        W = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
        varphi = 1 / (rng.random((N, F, T)) + 0.1)
        XX = (X[:, None] * X[None].conj()).transpose(2, 0, 1, 3)
        U = np.mean(varphi.transpose(1, 0, 2)[:, :, None, None, :] * XX[:, None], axis=-1)
        Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
        p = "n{}_".format(N)
        out[p + "W"], out[p + "U"], out[p + "Y"], out[p + "varphi"] = W, U, Y, varphi
        out[p + "ip2_out"] = update_by_ip2(W.copy(), U)
        out[p + "ip2_out_comb"] = update_by_ip2(W.copy(), U, pair_selector=combination_pair_selector)
        out[p + "ip2_out_add"] = update_by_ip2(W.copy(), U, flooring_fn=add)
        explicit = [(N - 1, 0), (0, N - 1)] if N > 2 else [(1, 0)]
        out[p + "ip2_pairs"] = np.array(explicit)
        out[p + "ip2_out_pairs"] = update_by_ip2(W.copy(), U, pair_selector=lambda n: explicit)
        W_keep = W.copy()
        out[p + "ip2_out_copy"] = update_by_ip2(W_keep, U, overwrite=False)
        assert np.array_equal(W_keep, W)
        out[p + "iss2_out"] = update_by_iss2(Y.copy(), varphi)
        out[p + "iss2_out_comb"] = update_by_iss2(Y.copy(), varphi,
                                                  pair_selector=combination_pair_selector)
        out[p + "iss2_out_bcast_add"] = update_by_iss2(Y.copy(), varphi[:, :1, :], flooring_fn=add)
        neg = [(-1, 0), (1, -N)] if N > 2 else [(-1, 0)]  # negative and descending indices
        out[p + "iss2_pairs"] = np.array(neg)
        out[p + "iss2_out_pairs"] = update_by_iss2(Y.copy(), varphi, pair_selector=lambda n: neg)
    save("pairwise_operators", **out)


def main():
    # --- GaussILRMA, IP1 ---
    run_ilrma("gilrma_ip1_n2", N=2, F=17, T=32, K=2, algo="IP", seed=0)            # KAT-1 of SURVEY 8c
    run_ilrma("gilrma_ip1_n3", N=3, F=20, T=37, K=16, algo="IP1", seed=10, gen=gen_mixture)
    run_ilrma("gilrma_ip1_n4", N=4, F=33, T=48, K=5, algo="IP", seed=20, gen=gen_mixture)
    run_ilrma("gilrma_ip1_n4_p1", N=4, F=18, T=40, K=4, algo="IP", seed=30, domain=1)
    run_ilrma("gilrma_ip1_n2_add", N=2, F=16, T=33, K=3, algo="IP", seed=40, flooring=("add", 1e-4))
    run_ilrma("gilrma_ip1_n2_nofloor", N=2, F=16, T=33, K=3, algo="IP", seed=41, flooring=("none", 0.0))
    run_ilrma("gilrma_ip1_n3_raw", N=3, F=19, T=34, K=4, algo="IP", seed=50,
              normalization=False, scale_restoration=False)
    # --- GaussILRMA, ISS1 ---
    run_ilrma("gilrma_iss1_n2", N=2, F=17, T=32, K=2, algo="ISS", seed=0)          # KAT-2
    run_ilrma("gilrma_iss1_n4", N=4, F=33, T=48, K=5, algo="ISS1", seed=20, gen=gen_mixture)
    run_ilrma("gilrma_iss1_n3_p1", N=3, F=18, T=40, K=4, algo="ISS", seed=31, domain=1)
    # --- pairwise updates (IP2 / ISS2); eigenvector phases make pre-projection-back filters and
    #     spectrograms comparable only up to a per-source phase
    run_ilrma("gilrma_ip2_n3", N=3, F=18, T=40, K=4, algo="IP2", seed=60, gen=gen_mixture)
    run_ilrma("gilrma_ip2_n2", N=2, F=17, T=32, K=2, algo="IP2", seed=61)
    run_ilrma("gilrma_iss2_n4", N=4, F=18, T=40, K=4, algo="ISS2", seed=62, gen=gen_mixture)
    run_ilrma("gilrma_iss2_n3", N=3, F=17, T=36, K=3, algo="ISS2", seed=63)
    run_iva("auxlap_ip2_n3", N=3, F=20, T=44, algo="IP2", contrast="laplace", seed=64, gen=gen_mixture)
    run_iva("auxlap_iss2_n4", N=4, F=20, T=44, algo="ISS2", contrast="laplace", seed=65, gen=gen_mixture)
    run_iva("auxgauss_ip2_n2", N=2, F=24, T=40, algo="IP2", contrast="gauss", seed=66)
    run_iva("auxgauss_iss2_n3", N=3, F=20, T=44, algo="ISS2", contrast="gauss", seed=67, gen=gen_mixture)
    # --- heavy-tailed source models (TILRMA, GGDILRMA)
    run_ilrma("tilrma_ip1_n3", N=3, F=18, T=40, K=4, algo="IP", seed=70, gen=gen_mixture, model=("t", 5.0))
    run_ilrma("tilrma_iss1_n2_p1", N=2, F=17, T=33, K=3, algo="ISS", seed=71, domain=1, model=("t", 100.0))
    run_ilrma("tilrma_ip2_n3", N=3, F=16, T=36, K=4, algo="IP2", seed=72, gen=gen_mixture, model=("t", 8.0))
    run_ilrma("ggdilrma_ip1_n3", N=3, F=18, T=40, K=4, algo="IP", seed=73, gen=gen_mixture, model=("ggd", 1.0))
    run_ilrma("ggdilrma_iss1_n2", N=2, F=17, T=33, K=3, algo="ISS", seed=74, model=("ggd", 1.5))
    run_ilrma("ggdilrma_iss2_n3_p1", N=3, F=16, T=36, K=4, algo="ISS2", seed=75, domain=1, gen=gen_mixture,
              model=("ggd", 0.7))
    # (round 6: the heavy-tailed ISS2 statistics come from the tuned covariance pass -- up to 4
    #  sources through identity filters, above that from the spectrogram itself)
    run_ilrma("tilrma_iss2_n4", N=4, F=17, T=40, K=4, algo="ISS2", seed=76, gen=gen_mixture, model=("t", 6.0))
    run_ilrma("ggdilrma_iss2_n4", N=4, F=16, T=44, K=3, algo="ISS2", seed=77, gen=gen_mixture,
              model=("ggd", 1.0))
    run_ilrma("tilrma_iss2_n6_p1", N=6, F=12, T=48, K=3, algo="ISS2", seed=78, domain=1, gen=gen_mixture,
              model=("t", 5.0))
    # --- ME source updates and partitioning (latent variables) ---
    run_ilrma("gilrma_me_ip1_n3", N=3, F=18, T=40, K=4, algo="IP", seed=90, gen=gen_mixture,
              source_algorithm="ME")
    run_ilrma("tilrma_me_iss1_n2", N=2, F=17, T=34, K=3, algo="ISS", seed=91, model=("t", 6.0),
              source_algorithm="ME")
    run_ilrma("gilrma_part_ip1_n3", N=3, F=18, T=40, K=6, algo="IP", seed=92, gen=gen_mixture,
              partitioning=True)
    run_ilrma("gilrma_part_iss1_n2_p1", N=2, F=17, T=34, K=5, algo="ISS", seed=93, domain=1,
              partitioning=True)
    run_ilrma("gilrma_part_me_ip2_n3", N=3, F=16, T=36, K=6, algo="IP2", seed=94, gen=gen_mixture,
              partitioning=True, source_algorithm="ME")
    run_ilrma("tilrma_part_ip1_n2", N=2, F=18, T=40, K=4, algo="IP", seed=95, model=("t", 5.0),
              partitioning=True)
    run_ilrma("ggdilrma_part_iss1_n3", N=3, F=16, T=36, K=6, algo="ISS", seed=96, gen=gen_mixture,
              model=("ggd", 1.3), partitioning=True)
    run_ilrma("tilrma_part_me_nonorm_n2", N=2, F=16, T=34, K=4, algo="IP", seed=97, model=("t", 4.0),
              partitioning=True, source_algorithm="ME", normalization=False)
    # --- AuxIVA ---
    run_iva("auxlap_ip1_n2", N=2, F=33, T=40, algo="IP", contrast="laplace", seed=0)
    run_iva("auxlap_ip1_n4", N=4, F=20, T=50, algo="IP1", contrast="laplace", seed=1, gen=gen_mixture)
    run_iva("auxlap_iss1_n2", N=2, F=33, T=40, algo="ISS", contrast="laplace", seed=0)
    run_iva("auxlap_iss1_n8", N=8, F=12, T=70, algo="ISS1", contrast="laplace", seed=2, gen=gen_mixture)
    run_iva("auxgauss_ip1_n3", N=3, F=24, T=45, algo="IP", contrast="gauss", seed=3, gen=gen_mixture)
    run_iva("auxgauss_iss1_n3", N=3, F=24, T=45, algo="ISS", contrast="gauss", seed=3, gen=gen_mixture)
    run_iva("auxlap_ip1_n2_raw", N=2, F=20, T=36, algo="IP", contrast="laplace", seed=4,
            scale_restoration=False)
    # config 1 of BASELINE.json (N=2, F=257, T=128, 10 it): known-answer scalars (KAT-3 / KAT-4)
    run_iva("kat_auxlap_ip1_config1", N=2, F=257, T=128, algo="IP", contrast="laplace", seed=0,
            keep_arrays=False)
    run_iva("kat_auxlap_iss1_config1", N=2, F=257, T=128, algo="ISS", contrast="laplace", seed=0,
            keep_arrays=False)
    # --- FastGaussMNMF ---
    run_mnmf("fmnmf_ip1_m3", M=3, F=17, T=32, K=2, seed=0)                          # KAT-5
    run_mnmf("fmnmf_ip1_m4", M=4, F=21, T=36, K=8, seed=5, gen=gen_mixture)
    run_mnmf("fmnmf_ip1_m3_n2", M=3, F=16, T=30, K=3, seed=6, n_sources=2)
    run_mnmf("fmnmf_ip1_m2_nonorm", M=2, F=16, T=30, K=3, seed=7, normalization=False)
    # --- FastGaussMNMF with the pairwise diagonaliser update (mnmf.py:1516-1633) ---
    run_mnmf("fmnmf_ip2_m2", M=2, F=16, T=30, K=3, seed=12, diag_algo="IP2")
    run_mnmf("fmnmf_ip2_m3", M=3, F=17, T=40, K=4, seed=13, gen=gen_mixture, diag_algo="IP2")
    run_mnmf("fmnmf_ip2_m4", M=4, F=15, T=36, K=5, seed=14, gen=gen_mixture, diag_algo="IP2")
    run_mnmf("fmnmf_ip2_m3_n2", M=3, F=16, T=32, K=3, seed=15, n_sources=2, diag_algo="IP2")
    run_mnmf("fmnmf_ip2_m4_comb", M=4, F=14, T=34, K=3, seed=16, diag_algo="IP2",
             pairs="combination")
    run_mnmf("fmnmf_ip2_m5", M=5, F=10, T=36, K=3, seed=17, diag_algo="IP2", n_iter=6)
    # shapes beyond 4 sources / channels (the general point-wise path of the device build)
    run_mnmf("fmnmf_ip1_m6_n3", M=6, F=12, T=40, K=4, seed=8, n_sources=3, gen=gen_mixture)
    run_mnmf("fmnmf_ip1_m5", M=5, F=10, T=36, K=3, seed=9)
    run_mnmf("fmnmf_ip1_m8_n2", M=8, F=8, T=48, K=2, seed=11, n_sources=2)
    # --- GaussMNMF (full-rank spatial covariance) ---
    run_gmnmf("gmnmf_m2", M=2, F=12, T=20, K=2, seed=80)
    run_gmnmf("gmnmf_m3", M=3, F=10, T=24, K=3, seed=81, gen=gen_mixture, spatial_init=True)
    run_gmnmf("gmnmf_m4_n3", M=4, F=9, T=22, K=4, seed=82, n_sources=3, spatial_init=True)
    run_gmnmf("gmnmf_m2_nonorm_add", M=2, F=10, T=18, K=2, seed=83, normalization=False,
              flooring=("add", 1e-6))
    run_gmnmf("gmnmf_part_m3", M=3, F=10, T=24, K=5, seed=84, gen=gen_mixture, spatial_init=True,
              partitioning=True)
    run_gmnmf("gmnmf_part_m2_n3", M=2, F=11, T=20, K=4, seed=85, n_sources=3, partitioning=True)
    # above 4 channels (the per-lane M x M kernels spill, but run)
    run_gmnmf("gmnmf_m5", M=5, F=9, T=26, K=3, seed=86, gen=gen_mixture, spatial_init=True, n_iter=6)
    run_gmnmf("gmnmf_m6_n3", M=6, F=7, T=30, K=2, seed=87, n_sources=3, n_iter=6)
    run_gmnmf("gmnmf_m8", M=8, F=5, T=36, K=2, seed=88, gen=gen_mixture, n_iter=4)
    # ... with the eigenvalue floor of to_psd ACTIVE (round 5: the packed per-point route of the device
    # build was pinned only against its own full-storage route there): eps = 0.3 against per-point
    # eigenvalues of order 1e-2 .. 1 clips most of R_ij, its inverse and the instantaneous covariance
    run_gmnmf("gmnmf_floor_m5", M=5, F=8, T=24, K=3, seed=140, gen=gen_mixture, spatial_init=True,
              flooring=("max", 0.3), n_iter=10)
    run_gmnmf("gmnmf_floor_m6_n3", M=6, F=6, T=28, K=2, seed=141, n_sources=3,
              flooring=("max", 0.3), n_iter=10)
    run_gmnmf("gmnmf_floor_m8", M=8, F=4, T=32, K=2, seed=142, gen=gen_mixture,
              flooring=("max", 0.3), n_iter=10)
    run_gmnmf("gmnmf_m7", M=7, F=5, T=34, K=2, seed=89, gen=gen_mixture, n_iter=4)
    run_eight_lane_operators()
    # --- IPA (iterative projection with adjustment, LQPQM solver) ---
    run_ipa_operators()
    run_ilrma("gilrma_ipa_n3", N=3, F=18, T=40, K=4, algo="IPA", seed=100, gen=gen_mixture)
    run_ilrma("gilrma_ipa_n2_p1", N=2, F=17, T=34, K=3, algo="IPA", seed=101, domain=1)
    run_ilrma("gilrma_ipa_part_n4", N=4, F=12, T=44, K=6, algo="IPA", seed=102, gen=gen_mixture,
              partitioning=True)
    run_ilrma("gilrma_ipa_newton8_n3", N=3, F=16, T=40, K=4, algo="IPA", seed=105, gen=gen_mixture,
              newton_iter=8)
    run_iva("auxlap_ipa_n3", N=3, F=20, T=44, algo="IPA", contrast="laplace", seed=103, gen=gen_mixture)
    run_iva("auxgauss_ipa_n2", N=2, F=24, T=40, algo="IPA", contrast="gauss", seed=104)
    # --- scale restoration by the minimal distortion principle, projection-back normalisation ---
    run_ilrma("gilrma_mdp_ip1_n3", N=3, F=16, T=36, K=3, algo="IP", seed=110, gen=gen_mixture,
              scale_restoration="minimal_distortion_principle")
    run_ilrma("gilrma_mdp_iss1_n2", N=2, F=16, T=34, K=3, algo="ISS", seed=111,
              scale_restoration="minimal_distortion_principle")
    run_ilrma("gilrma_pbnorm_ip1_n3", N=3, F=16, T=36, K=3, algo="IP", seed=112, gen=gen_mixture,
              normalization="projection_back")
    run_ilrma("gilrma_pbnorm_iss1_n2_p1", N=2, F=16, T=34, K=3, algo="ISS", seed=113, domain=1,
              normalization="projection_back")
    run_iva("auxlap_mdp_ip1_n3", N=3, F=20, T=44, algo="IP", contrast="laplace", seed=114,
            gen=gen_mixture, scale_restoration="minimal_distortion_principle")
    run_iva("auxlap_mdp_iss1_n2", N=2, F=20, T=40, algo="ISS", contrast="laplace", seed=115,
            scale_restoration="minimal_distortion_principle")
    # --- boundary: user contrast closures, all-channel scale restoration ---
    run_generic_auxiva("auxgeneric_ip1_n3", N=3, F=20, T=44, algo="IP", seed=120, gen=gen_mixture)
    run_generic_auxiva("auxgeneric_iss1_n2", N=2, F=24, T=40, algo="ISS", seed=121)
    run_generic_auxiva("auxgeneric_ip2_n3", N=3, F=18, T=40, algo="IP2", seed=122, gen=gen_mixture)
    run_all_channel_restoration()
    # --- rng-drawn initial state (nothing injected): pins the draw order of _init_nmf & friends ---
    run_rng_init("rnginit_gilrma_n3", cls="ilrma", seed=130, N=3, F=14, T=30, K=4,
                 spatial_algorithm="IP")
    run_rng_init("rnginit_gilrma_part_n3", cls="ilrma", seed=131, N=3, F=14, T=30, K=5,
                 spatial_algorithm="IP", partitioning=True)
    run_rng_init("rnginit_gilrma_part_iss_n2", cls="ilrma", seed=132, N=2, F=15, T=28, K=4,
                 spatial_algorithm="ISS", partitioning=True)
    run_rng_init("rnginit_tilrma_part_n2", cls="ilrma", seed=133, N=2, F=13, T=32, K=4,
                 spatial_algorithm="IP", partitioning=True, model=("t", 5.0))
    run_rng_init("rnginit_fmnmf_m3", cls="fmnmf", seed=134, N=3, F=12, T=28, K=3)
    run_rng_init("rnginit_gmnmf_m2", cls="gmnmf", seed=135, N=2, F=9, T=20, K=3)
    run_rng_init("rnginit_gmnmf_part_m2", cls="gmnmf", seed=136, N=2, F=9, T=20, K=4,
                 partitioning=True)
    # --- arbitrary flooring callables ---
    run_custom_floor("customfloor_gilrma_ip1_n3", kind="ilrma", seed=150, N=3, F=14, T=30,
                     spatial_algorithm="IP")
    run_custom_floor("customfloor_gilrma_iss1_n2", kind="ilrma", seed=151, N=2, F=15, T=28,
                     spatial_algorithm="ISS")
    run_custom_floor("customfloor_auxlap_ip1_n3", kind="iva", seed=152, N=3, F=14, T=30,
                     spatial_algorithm="IP")
    run_custom_floor("customfloor_auxlap_iss1_n2", kind="iva", seed=153, N=2, F=15, T=28,
                     spatial_algorithm="ISS")
    run_custom_floor("customfloor_fmnmf_m3", kind="fmnmf", seed=154, N=3, F=12, T=28, K=3)
    # ... on the pairwise updates (two denominators per pair), under partitioning, AuxGaussIVA
    run_custom_floor("customfloor_gilrma_ip2_n3", kind="ilrma", seed=155, N=3, F=14, T=30,
                     spatial_algorithm="IP2")
    run_custom_floor("customfloor_gilrma_iss2_n4", kind="ilrma", seed=156, N=4, F=12, T=32,
                     spatial_algorithm="ISS2")
    run_custom_floor("customfloor_gilrma_part_ip1_n3", kind="ilrma", seed=157, N=3, F=14, T=30, K=5,
                     spatial_algorithm="IP", partitioning=True)
    run_custom_floor("customfloor_gilrma_part_iss1_n2", kind="ilrma", seed=158, N=2, F=15, T=28, K=4,
                     spatial_algorithm="ISS", partitioning=True)
    run_custom_floor("customfloor_auxlap_ip2_n3", kind="iva", seed=159, N=3, F=14, T=30,
                     spatial_algorithm="IP2")
    run_custom_floor("customfloor_auxlap_iss2_n3", kind="iva", seed=165, N=3, F=14, T=30,
                     spatial_algorithm="ISS2")
    run_custom_floor("customfloor_auxgauss_ip1_n3", kind="gaussiva", seed=166, N=3, F=14, T=30,
                     spatial_algorithm="IP")
    run_custom_floor("customfloor_auxgauss_iss1_n2", kind="gaussiva", seed=167, N=2, F=15, T=28,
                     spatial_algorithm="ISS")
    run_custom_floor("customfloor_auxgauss_ip2_n3", kind="gaussiva", seed=168, N=3, F=14, T=30,
                     spatial_algorithm="IP2")
    run_custom_floor("customfloor_tilrma_ip2_n3", kind="tilrma", seed=169, N=3, F=14, T=30, dof=5.0,
                     spatial_algorithm="IP2")
    run_custom_floor("customfloor_tilrma_iss2_n3", kind="tilrma", seed=170, N=3, F=14, T=30, dof=5.0,
                     spatial_algorithm="ISS2")
    run_custom_floor("customfloor_ggdilrma_ip1_n3", kind="ggdilrma", seed=171, N=3, F=14, T=30, beta=1.0,
                     spatial_algorithm="IP")
    run_custom_floor("customfloor_ggdilrma_iss1_n2", kind="ggdilrma", seed=172, N=2, F=15, T=28, beta=1.5,
                     spatial_algorithm="ISS")
    run_custom_floor("customfloor_ggdilrma_ip2_n3", kind="ggdilrma", seed=173, N=3, F=14, T=30, beta=0.7,
                     spatial_algorithm="IP2")
    run_custom_floor("customfloor_ggdilrma_iss2_n3", kind="ggdilrma", seed=174, N=3, F=14, T=30, beta=1.0,
                     spatial_algorithm="ISS2")
    # --- more than 8 sources (the reference takes n_sources from input.shape without a limit) ---
    run_ilrma("gilrma_ip1_n10", N=10, F=12, T=80, K=3, algo="IP", seed=160, gen=gen_mixture, n_iter=6)
    run_ilrma("gilrma_iss1_n9_p1", N=9, F=10, T=72, K=2, algo="ISS", seed=161, domain=1, n_iter=6)
    run_iva("auxlap_iss1_n12", N=12, F=8, T=96, algo="ISS", contrast="laplace", seed=162,
            gen=gen_mixture, n_iter=6)
    run_iva("auxlap_ip1_n9", N=9, F=10, T=80, algo="IP", contrast="laplace", seed=163, n_iter=6)
    run_iva("auxgauss_ip1_n16_mdp", N=16, F=6, T=128, algo="IP", contrast="gauss", seed=164,
            gen=gen_mixture, n_iter=4, scale_restoration="minimal_distortion_principle")
    run_ilrma("gilrma_ip2_n9", N=9, F=8, T=72, K=2, algo="IP2", seed=169, gen=gen_mixture, n_iter=4)
    run_ilrma("gilrma_iss2_n10", N=10, F=8, T=80, K=3, algo="ISS2", seed=170, n_iter=4)
    run_iva("auxlap_ip2_n9", N=9, F=8, T=72, algo="IP2", contrast="laplace", seed=171, n_iter=4)
    run_iva("auxlap_iss2_n12", N=12, F=6, T=96, algo="ISS2", contrast="laplace", seed=172,
            gen=gen_mixture, n_iter=4)
    # (round 6: IPA with the source count at run time, ipa_rt.hip)
    run_ilrma("gilrma_ipa_n9", N=9, F=7, T=108, K=2, algo="IPA", seed=175, gen=gen_mixture, n_iter=4)
    run_ilrma("gilrma_ipa_n12_add", N=12, F=5, T=132, K=2, algo="IPA", seed=176, gen=gen_mixture, n_iter=3,
              flooring=("add", 1e-9), newton_iter=3)
    run_iva("auxlap_ipa_n10", N=10, F=6, T=110, algo="IPA", contrast="laplace", seed=177,
            gen=gen_mixture, n_iter=4)
    # --- 3 / 4 sources with at least 16 frames per source (round 6): the device build runs these
    #     ISS / ISS2 / IPA iterations through the filters the updates imply (W <- G W, statistics
    #     W U W^H) -- reference vectors for that route, ISS1 on a batch-sized number of bins
    run_ilrma("gilrma_iss1_n4_t80", N=4, F=33, T=80, K=5, algo="ISS1", seed=180, gen=gen_mixture)
    run_ilrma("gilrma_iss2_n4_t72", N=4, F=18, T=72, K=4, algo="ISS2", seed=181, gen=gen_mixture)
    run_ilrma("gilrma_iss2_n3_t64", N=3, F=17, T=64, K=3, algo="ISS2", seed=182)
    run_ilrma("gilrma_ipa_n3_t56", N=3, F=18, T=56, K=4, algo="IPA", seed=183, gen=gen_mixture)
    run_ilrma("gilrma_ipa_n4_t72", N=4, F=16, T=72, K=4, algo="IPA", seed=184, gen=gen_mixture)
    run_iva("auxlap_iss2_n4_t72", N=4, F=20, T=72, algo="ISS2", contrast="laplace", seed=185,
            gen=gen_mixture)
    run_iva("auxlap_ipa_n3_t60", N=3, F=20, T=60, algo="IPA", contrast="laplace", seed=186,
            gen=gen_mixture)
    run_iva("auxgauss_ipa_n4_t70", N=4, F=16, T=70, algo="IPA", contrast="gauss", seed=187)
    # --- operators ---
    run_operators()
    run_pairwise_operators()
    run_lqpqm_singular()
    run_psd_custom_floor()


if __name__ == "__main__":
    main()
