"""GPU sweep over the option cross-product the reference's own package tests walk
(tests/package/bss/test_ilrma.py, test_iva.py, test_mnmf.py: spatial_algorithm x source_algorithm x
normalization x scale_restoration x partitioning, callbacks None / function / list), on small random
mixtures.  The reference tests only check shapes and types; here every combination is also compared
with the CPU oracle after three iterations.
"""

import itertools

import numpy as np
import pytest

from conftest import rel_err, rel_err_up_to_phase

pytestmark = pytest.mark.gpu

TOL = 1e-8
n_iter = 3
parameters_spatial_algorithm = ["IP", "IP1", "IP2", "ISS", "ISS1", "ISS2", "IPA"]
parameters_source_algorithm = ["MM", "ME"]
parameters_scale_restoration = [True, False, "projection_back", "minimal_distortion_principle"]
parameters_normalization_latent = [True, False, "power"]
parameters_normalization_wo_latent = [True, False, "power", "projection_back"]
parameters_callbacks = ["none", "function", "list"]


def _mixture(seed, N, F, T):
    from ssspy_amd.utils.dataset import nmf_mixture

    return nmf_mixture(seed, N, F, T)


def _callbacks(kind, log):
    def dummy_function(method):
        log.append(type(method).__name__)

    class DummyCallback:
        def __call__(self, method):
            log.append("cb")

    return {"none": None, "function": dummy_function, "list": [DummyCallback(), dummy_function]}[kind]


def _compare_ilrma(m, ref, Y, Yr, algo):
    pairwise = algo in ("IP2", "ISS2")
    err = rel_err_up_to_phase(Y, Yr, "output") if pairwise and not m.scale_restoration else rel_err(Y, Yr)
    assert err < TOL
    assert rel_err(m.basis, ref.basis) < TOL and rel_err(m.activation, ref.activation) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-9)
    assert (m.demix_filter is None) == (ref.demix_filter is None)


def _ilrma_cases(models):
    for model in models:
        for part in (False, True):
            norms = parameters_normalization_latent if part else parameters_normalization_wo_latent
            for algo, src, norm, scale in itertools.product(
                    parameters_spatial_algorithm, parameters_source_algorithm, norms,
                    parameters_scale_restoration):
                if model[0] != "gauss" and algo == "IPA":
                    continue  # ValueError in the reference
                if model[0] == "ggd" and src == "ME":
                    continue  # GGD-ILRMA is MM only
                yield model, part, algo, src, norm, scale


ILRMA_SWEEP = list(_ilrma_cases([("gauss", None), ("t", 100.0), ("ggd", 1.5)]))


@pytest.mark.parametrize("chunk", range(8))
def test_ilrma_option_sweep(chunk):
    """1/8 of the (model, partitioning, spatial, source, normalization, scale restoration) product
    per test so a failure names a small group."""
    import functools

    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GGDILRMA, TILRMA, GaussILRMA

    cases = ILRMA_SWEEP[chunk::8]
    assert len(ILRMA_SWEEP) > 600
    for idx, (model, part, algo, src, norm, scale) in enumerate(cases):
        N, F, T, K = (3, 9, 25, 4) if idx % 2 else (2, 8, 24, 3)
        X = _mixture(300 + idx, N, F, T)
        rng = np.random.default_rng(idx)
        lead = () if part else (N,)
        init = dict(basis=rng.random(lead + (F, K)), activation=rng.random(lead + (K, T)))
        if part:
            Z = rng.random((N, K))
            init["latent"] = Z / Z.sum(axis=0)
        log = []
        cls = {"gauss": GaussILRMA, "t": functools.partial(TILRMA, dof=model[1]),
               "ggd": functools.partial(GGDILRMA, beta=model[1])}[model[0]]
        m = cls(n_basis=K, spatial_algorithm=algo, source_algorithm=src, partitioning=part,
                normalization=norm, scale_restoration=scale,
                callbacks=_callbacks(parameters_callbacks[idx % 3], log))
        Y = m(X, n_iter=n_iter, **init)
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, source_algorithm=src,
                               partitioning=part, normalization=norm, scale_restoration=scale,
                               model=model)
        Yr = ref.run(X, n_iter=n_iter, **init)
        tag = (model, part, algo, src, norm, scale)
        try:
            _compare_ilrma(m, ref, Y, Yr, algo)
        except AssertionError as e:
            raise AssertionError("{}: {}".format(tag, e))
        assert Y.shape == X.shape and type(m.loss[-1]) is float
        expect_calls = {0: 0, 1: n_iter + 1, 2: 2 * (n_iter + 1)}[idx % 3]
        assert len(log) == expect_calls


def test_ilrma_projection_back_normalization_rejects_partitioning():
    from ssspy_amd.bss.ilrma import GaussILRMA

    X = _mixture(1, 2, 8, 24)
    with pytest.raises(NotImplementedError):
        GaussILRMA(n_basis=2, partitioning=True, normalization="projection_back")(X, n_iter=1)


@pytest.mark.parametrize("contrast", ["laplace", "gauss"])
def test_aux_iva_option_sweep(contrast):
    from oracle.iva import AuxIVAOracle
    from ssspy_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA

    cls = AuxLaplaceIVA if contrast == "laplace" else AuxGaussIVA
    for idx, (algo, scale) in enumerate(itertools.product(parameters_spatial_algorithm,
                                                          parameters_scale_restoration)):
        N, F, T = (3, 9, 30) if idx % 2 else (2, 8, 28)
        X = _mixture(500 + idx, N, F, T)
        m = cls(spatial_algorithm=algo, scale_restoration=scale)
        Y = m(X, n_iter=n_iter)
        ref = AuxIVAOracle(spatial_algorithm=algo, contrast=contrast, scale_restoration=scale)
        Yr = ref.run(X, n_iter=n_iter)
        pairwise = algo in ("IP2", "ISS2")
        err = rel_err_up_to_phase(Y, Yr, "output") if pairwise and not scale else rel_err(Y, Yr)
        assert err < TOL, (algo, scale, err)
        np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-9, err_msg=str((algo, scale)))


@pytest.mark.parametrize("partitioning", [False, True])
def test_gauss_mnmf_option_sweep(partitioning):
    from oracle.gmnmf import GaussMNMFOracle
    from ssspy_amd.bss.mnmf import GaussMNMF

    for idx, (n_sources, normalization) in enumerate(itertools.product([None, 3], [True, False])):
        M, F, T, K = 2, 7, 22, 3
        X = _mixture(700 + idx, M, F, T)
        N = M if n_sources is None else n_sources
        rng = np.random.default_rng(idx)
        lead = () if partitioning else (N,)
        init = dict(basis=rng.random(lead + (F, K)), activation=rng.random(lead + (K, T)))
        if partitioning:
            Z = rng.random((N, K))
            init["latent"] = Z / Z.sum(axis=0)
        m = GaussMNMF(n_basis=K, n_sources=n_sources, partitioning=partitioning,
                      normalization=normalization)
        Y = m(X, n_iter=n_iter, **init)
        ref = GaussMNMFOracle(n_basis=K, n_sources=n_sources, partitioning=partitioning,
                              normalization=normalization)
        Yr = ref.run(X, n_iter=n_iter, **init)
        assert Y.shape == (N, F, T)
        assert rel_err(Y, Yr) < 1e-7 and rel_err(m.spatial, ref.spatial) < 1e-7
        np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-8)


def test_fast_gauss_mnmf_option_sweep():
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF

    for idx, (n_sources, normalization) in enumerate(itertools.product([None, 2], [True, False])):
        M, F, T, K = 3, 7, 22, 3
        X = _mixture(800 + idx, M, F, T)
        N = M if n_sources is None else n_sources
        rng = np.random.default_rng(idx)
        init = dict(basis=rng.random((N, F, K)), activation=rng.random((N, K, T)),
                    spatial=rng.random((F, N, M)))
        m = FastGaussMNMF(n_basis=K, n_sources=n_sources, normalization=normalization)
        Y = m(X, n_iter=n_iter, **{k: v.copy() for k, v in init.items()})
        ref = FastGaussMNMFOracle(n_basis=K, n_sources=n_sources, normalization=normalization)
        Yr = ref.run(X, n_iter=n_iter, **init)
        assert rel_err(Y, Yr) < 1e-7 and rel_err(m.diagonalizer, ref.diagonalizer) < TOL
        np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-9)


@pytest.mark.parametrize("shape", [(2, 1, 2, 1), (2, 5, 7, 1), (3, 16, 16, 16), (4, 17, 18, 16),
                                   (4, 64, 1024, 16), (2, 1025, 2, 3), (4, 15, 510, 9), (8, 3, 33, 2),
                                   (4, 129, 66, 64), (4, 70, 513, 16), (3, 33, 31, 5), (2, 20, 3, 2),
                                   (4, 16, 17, 16)])
@pytest.mark.parametrize("algo", ["IP", "ISS"])
def test_gauss_ilrma_edge_shapes(shape, algo):
    """Tiny, ragged and lopsided shapes: one bin, two frames, tiles with a single valid row or
    column, n_basis 1 and 64, odd and even n_frames."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA

    N, F, T, K = shape
    X = _mixture(900 + F, N, F, T)
    rng = np.random.default_rng(F * T)
    basis, act = rng.random((N, F, K)), rng.random((N, K, T))
    m = GaussILRMA(n_basis=K, spatial_algorithm=algo)
    Y = m(X, n_iter=2, basis=basis, activation=act)
    ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
    Yr = ref.run(X, n_iter=2, basis=basis, activation=act)
    assert rel_err(Y, Yr) < TOL, shape
    np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-9)
