"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors made from the
reference and against the CPU oracle on the same seeded inputs.

Tolerances: north_star asks for 1e-4 relative (fp64) on filters and spectrograms; the kernels
are held to 1e-8 here on 10-iteration runs (measured round-off amplification of IP over 10
iterations is ~1e3), and losses to 1e-9 relative.
"""

import functools
import warnings

import numpy as np
import pytest

from conftest import load_golden, rel_err, rel_err_up_to_phase


def pair_err(a, b, name, N):
    """Pairwise updates against a reference / oracle result.  From 3 sources on WITH the phases
    np.linalg.eigh leaves in the 2 x 2 eigenvectors (csrc/eigh2.hpp restates LAPACK's convention,
    round 6); with 2 sources the default selectors visit the one pair twice and the second visit's
    phase is LAPACK's reading of rounding noise: up to a phase per row there."""
    return rel_err(a, b) if N >= 3 else rel_err_up_to_phase(a, b, name)
from conftest import option as _option

from ssspy_amd import _routes

pytestmark = pytest.mark.gpu

TOL = 1e-8
LOSS_RTOL = 1e-9

ILRMA_CASES = [
    "gilrma_ip1_n2", "gilrma_ip1_n3", "gilrma_ip1_n4", "gilrma_ip1_n4_p1", "gilrma_ip1_n2_add",
    "gilrma_ip1_n2_nofloor", "gilrma_ip1_n3_raw", "gilrma_iss1_n2", "gilrma_iss1_n4",
    "gilrma_iss1_n3_p1", "gilrma_ip2_n3", "gilrma_ip2_n2", "gilrma_iss2_n4", "gilrma_iss2_n3",
    "tilrma_ip1_n3", "tilrma_iss1_n2_p1", "tilrma_ip2_n3", "ggdilrma_ip1_n3", "ggdilrma_iss1_n2",
    "ggdilrma_iss2_n3_p1", "tilrma_iss2_n4", "ggdilrma_iss2_n4", "tilrma_iss2_n6_p1", "gilrma_me_ip1_n3", "tilrma_me_iss1_n2", "gilrma_part_ip1_n3",
    "gilrma_part_iss1_n2_p1", "gilrma_part_me_ip2_n3", "tilrma_part_ip1_n2", "ggdilrma_part_iss1_n3",
    "tilrma_part_me_nonorm_n2", "gilrma_ipa_n3", "gilrma_ipa_n2_p1", "gilrma_ipa_part_n4",
    "gilrma_ipa_newton8_n3",
    "gilrma_mdp_ip1_n3", "gilrma_mdp_iss1_n2", "gilrma_pbnorm_ip1_n3", "gilrma_pbnorm_iss1_n2_p1",
    "gilrma_ip1_n10", "gilrma_iss1_n9_p1",  # above 8 sources: the run-time-N kernels (wide_n.hip)
    "gilrma_ip2_n9", "gilrma_iss2_n10", "gilrma_ipa_n9", "gilrma_ipa_n12_add",
    # 3 / 4 sources, >= 16 frames per source (round 6: the implied-filter route of the device build)
    "gilrma_iss1_n4_t80", "gilrma_iss2_n4_t72", "gilrma_iss2_n3_t64", "gilrma_ipa_n3_t56",
    "gilrma_ipa_n4_t72",
]
IVA_CASES = [
    "auxlap_ip1_n2", "auxlap_ip1_n4", "auxlap_iss1_n2", "auxlap_iss1_n8", "auxgauss_ip1_n3",
    "auxgauss_iss1_n3", "auxlap_ip1_n2_raw", "auxlap_ip2_n3", "auxlap_iss2_n4", "auxgauss_ip2_n2",
    "auxgauss_iss2_n3", "auxlap_ipa_n3", "auxgauss_ipa_n2", "auxlap_mdp_ip1_n3",
    "auxlap_mdp_iss1_n2",
    "auxlap_iss1_n12", "auxlap_ip1_n9", "auxgauss_ip1_n16_mdp",  # above 8 sources (wide_n.hip)
    "auxlap_ip2_n9", "auxlap_iss2_n12", "auxlap_ipa_n10",
    "auxlap_iss2_n4_t72", "auxlap_ipa_n3_t60", "auxgauss_ipa_n4_t70",  # (round 6, see above)
]


def _flooring_fn(g):
    from ssspy_amd.special.flooring import add_flooring, max_flooring

    kind, eps = str(g["meta_floor_kind"]), float(g["meta_floor_eps"])
    if kind == "max":
        return functools.partial(max_flooring, eps=eps)
    if kind == "add":
        return functools.partial(add_flooring, eps=eps)
    return None


class Snap:
    def __init__(self, names):
        self.names, self.count, self.store = names, -1, {}

    def __call__(self, m):
        self.count += 1
        if self.count in (1, 2, 10):
            for name in self.names:
                v = getattr(m, name, None)
                if v is not None:
                    self.store["it{}_{}".format(self.count, name)] = np.array(v, copy=True)


def _compare_snapshots(g, snap):
    pairwise = "meta_algo" in g and str(g["meta_algo"]) in ("IP2", "ISS2")
    checked = 0
    for key, value in snap.store.items():
        assert key in g, key
        name = key.split("_", 1)[1]
        if pairwise and name in ("demix_filter", "output") and g["X"].shape[0] == 2:
            # two sources: the default selectors visit the pair twice, and the second visit's
            # eigenvector phase is LAPACK's reading of rounding noise (pair_phase_check.py)
            err = rel_err_up_to_phase(value, g[key], name)
        else:
            # (pairwise updates from 3 sources on: the reference's snapshots WITH the phases its
            #  np.linalg.eigh leaves in the unrestored filters / outputs -- csrc/eigh2.hpp, round 6;
            #  compared up to a phase per row before)
            err = rel_err(value, g[key])
            if pairwise and name in ("demix_filter", "output"):
                # (a phase is as well determined as the off-diagonal entry it is read from: 1e-7
                #  after ten GGD iterations where the moduli agree to 1e-9)
                assert err < 1e-6, "{}: {}".format(key, err)
                err = rel_err_up_to_phase(value, g[key], name)
        assert err < TOL, "{}: {}".format(key, err)
        checked += 1
    assert checked > 0


# ------------------------------------------------------------------------------- operators
@pytest.mark.parametrize("N", [2, 3, 4, 8])
def test_operators_against_golden(N):
    from ssspy_amd.algorithm import projection_back
    from ssspy_amd.bss._update_spatial_model import update_by_ip1, update_by_iss1
    from ssspy_amd.special.flooring import add_flooring

    g = load_golden("operators")
    p = lambda s: g[s.format(N)]  # noqa: E731
    W = p("ip1_n{}_W").copy()
    out = update_by_ip1(W, p("ip1_n{}_U"))
    assert out is W  # overwrite=True aliases, as in the reference
    assert rel_err(out, p("ip1_n{}_out")) < 1e-11
    out = update_by_ip1(p("ip1_n{}_W"), p("ip1_n{}_U"),
                        flooring_fn=functools.partial(add_flooring, eps=1e-3), overwrite=False)
    assert rel_err(out, p("ip1_n{}_out_add")) < 1e-11
    assert rel_err(update_by_iss1(p("iss1_n{}_Y"), p("iss1_n{}_varphi")), p("iss1_n{}_out")) < 1e-11
    assert rel_err(update_by_iss1(p("iss1_n{}_Y"), p("iss1_n{}_varphi")[:, :1, :]),
                   p("iss1_n{}_out_bcast")) < 1e-11
    assert rel_err(projection_back(p("ip1_n{}_W"), reference_id=1), p("pb_n{}_filter")) < 1e-11
    assert rel_err(projection_back(p("iss1_n{}_Y"), reference=p("pb_n{}_X"), reference_id=0),
                   p("pb_n{}_output")) < 1e-11


@pytest.mark.parametrize("N", [2, 3, 4, 5])
def test_ipa_operator_against_golden(N):
    """update_by_ipa (LQPQM solver per bin) against the reference: default, no normalisation with
    three Newton steps, broadcast weights with an additive floor."""
    from ssspy_amd.bss._update_spatial_model import update_by_ipa
    from ssspy_amd.special.flooring import add_flooring

    g = load_golden("ipa_operators")
    Y, varphi = g["n{}_Y".format(N)], g["n{}_varphi".format(N)]
    Y0 = Y.copy()
    assert rel_err(update_by_ipa(Y, varphi), g["n{}_out".format(N)]) < 1e-10
    assert np.array_equal(Y, Y0)
    assert rel_err(update_by_ipa(Y, varphi, normalization=False, max_iter=3),
                   g["n{}_out_nonorm_it3".format(N)]) < 1e-10
    out = update_by_ipa(Y, varphi[:, :1, :], flooring_fn=functools.partial(add_flooring, eps=1e-4))
    assert rel_err(out, g["n{}_out_bcast_add".format(N)]) < 1e-10
    # twelve steps allowed: the loop stops as soon as every bin of the call has converged
    assert rel_err(update_by_ipa(Y, varphi, max_iter=12), g["n{}_out_it12".format(N)]) < 1e-10


@pytest.mark.parametrize("N", [7, 8])
def test_ipa_and_hermitian_operators_against_reference_vectors_at_7_and_8(N):
    """Round 5: the kernels with a bin / matrix on 8 lanes (IPA at 8 sources, the Hermitian
    operators from 7 x 7; the lane-per-bin IPA at 7 sources) against vectors generated by the
    reference itself (fixture eight_lane_operators): update_by_ipa in four settings, sqrtmh,
    invsqrtmh (also with an acting floor), to_psd (max floor default / acting, add floor),
    gmeanmh and the generalised eigh of the three types (eigenvalues; eigenvectors through the
    defining equation, their phase being the decomposition's own)."""
    from ssspy_amd.bss._update_spatial_model import update_by_ipa
    from ssspy_amd.linalg import eigh, gmeanmh, invsqrtmh, sqrtmh
    from ssspy_amd.special.flooring import add_flooring, max_flooring
    from ssspy_amd.special.psd import to_psd

    g = load_golden("eight_lane_operators")
    Y, varphi = g["ipa{}_Y".format(N)], g["ipa{}_varphi".format(N)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert rel_err(update_by_ipa(Y, varphi), g["ipa{}_out".format(N)]) < 1e-9
        assert rel_err(update_by_ipa(Y, varphi, normalization=False, max_iter=3),
                       g["ipa{}_out_nonorm_it3".format(N)]) < 1e-7
        out = update_by_ipa(Y, varphi[:, :1, :],
                            flooring_fn=functools.partial(add_flooring, eps=1e-4))
        assert rel_err(out, g["ipa{}_out_bcast_add".format(N)]) < 1e-9
        assert rel_err(update_by_ipa(Y, varphi, max_iter=12), g["ipa{}_out_it12".format(N)]) < 1e-9
    M = N
    k = "m{}_".format(M)
    A, B, H = g[k + "A"], g[k + "B"], g[k + "H"]
    assert rel_err(sqrtmh(A), g[k + "sqrtmh"]) < 1e-11
    assert rel_err(invsqrtmh(A), g[k + "invsqrtmh"]) < 1e-10
    assert rel_err(invsqrtmh(A, flooring_fn=functools.partial(max_flooring, eps=0.6)),
                   g[k + "invsqrtmh_floor"]) < 1e-10
    assert rel_err(to_psd(H), g[k + "to_psd"]) < 1e-11
    assert rel_err(to_psd(H, flooring_fn=functools.partial(max_flooring, eps=0.5)),
                   g[k + "to_psd_floor"]) < 1e-11
    assert rel_err(to_psd(H, flooring_fn=functools.partial(add_flooring, eps=0.25)),
                   g[k + "to_psd_add"]) < 1e-11
    for t in (1, 2, 3):
        assert rel_err(gmeanmh(A, B, type=t), g[k + "gmeanmh{}".format(t)]) < 1e-10
        lamb, z = eigh(A, B, type=t)
        np.testing.assert_allclose(lamb, g[k + "eigh{}_lamb".format(t)], rtol=1e-10, atol=1e-12)
        zr = g[k + "eigh{}_z".format(t)]
        proj = lambda v: v[..., :, None, :] * v[..., None, :, :].conj()  # noqa: E731
        assert rel_err(proj(z), proj(zr)) < 1e-8  # (z_k z_k^H: the phase of z_k drops out)


@pytest.mark.parametrize("newton_iter", [1, 6, 40])
def test_ipa_newton_step_count_is_per_mixture(newton_iter):
    """The reference stops its joint Newton loop when every bin of THE mixture has converged
    (ssspy/linalg/lqpqm.py:196-213): in a batch each mixture keeps its own step count, so every
    element equals its single-mixture run, and element 0 the oracle."""
    import warnings

    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, N, F, T, K = 4, 3, 33, 60, 4
    X = np.stack([nmf_mixture(810 + 7 * b, N, F, T) for b in range(B)])
    rng = np.random.default_rng(21)
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (the reference's "did not converge" at small step counts)
        m = GaussILRMA(n_basis=K, spatial_algorithm="IPA", newton_iter=newton_iter)
        Y = m(X, n_iter=3, basis=basis, activation=act)
        for b in range(B):
            s1 = GaussILRMA(n_basis=K, spatial_algorithm="IPA", newton_iter=newton_iter)
            Yb = s1(X[b], n_iter=3, basis=basis[b], activation=act[b])
            assert rel_err(Y[b], Yb) < 1e-12, b
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm="IPA")
        ref.newton_iter = newton_iter
        Yr = ref.run(X[0], n_iter=3, basis=basis[0], activation=act[0])
    assert rel_err(Y[0], Yr) < TOL


@pytest.mark.parametrize("N", [2, 3, 4, 6, 8])
def test_ipa_chained_sweep_against_oracle(N):
    """Round 5: one weighted covariance, the N source steps chained on the per-bin statistics
    (V_m <- G V_m G^H) and one Y <- G Y (ssspy_ipa_sweep) against the oracle's literal per-source
    passes (covariance -> update matrix -> Y <- G Y, N times: ssspy/bss/_update_spatial_model.py:
    398-513), with different Newton step counts."""
    from oracle.ipa import update_by_ipa as oracle_ipa
    from ssspy_amd.bss._update_spatial_model import update_by_ipa

    rng = np.random.default_rng(40 + N)
    F, T = 9, 50
    Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    varphi = 1.0 / (rng.random((N, F, T)) + 0.05)
    for kw in (dict(), dict(normalization=False, max_iter=4), dict(max_iter=9)):
        a = update_by_ipa(Y, varphi, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = oracle_ipa(Y, varphi, **kw)
        assert rel_err(a, ref) < 1e-10


@pytest.mark.parametrize("N", [5, 6, 7, 8])
def test_ipa_above_four_sources_against_oracle(N):
    """The IPA source step at 5-7 sources (a lane per bin) and at 8 (a bin on 8 lanes, ipa_rows.hip)
    against the oracle: the three floors (the max floor made to act on a fifth of the statistics, which sends
    whole bins down the eigen route), both normalisations, Newton probe / apply, and a ragged last
    block of bins."""
    from oracle.ipa import update_by_ipa as oracle_ipa
    from ssspy_amd.bss._update_spatial_model import update_by_ipa
    from ssspy_amd.special.flooring import add_flooring, max_flooring

    rng = np.random.default_rng(70 + N)
    F, T = 37, 60  # (37 bins: one full block of 32 and a ragged one)
    Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    varphi = 1.0 / (rng.random((N, F, T)) + 0.05)
    # For the max floor that ACTS: instantaneous mixtures (correlated channels), scaled so that a floor
    # above the smallest eigenvalue of a tenth of the (bin, weight set) statistics stays far below
    # the terms the algorithm compares with floor(0) (||v||, phi |v~|^2, |f|): where those sit at the
    # threshold the reference's own result jumps -- with every phi |v~|^2 masked its cubic has a double
    # root at 1 and "lambda > 1" is decided by rounding (1e-3 between any two implementations).  This
    # regime moves by 1e-10 under a 1e-13 perturbation of the input (checked with the oracle).
    A = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
    Ymix = 0.1 * np.einsum("fnm,mft->nft", A, Y)
    YY = Ymix[:, None] * Ymix[None, :].conj()
    U = np.mean(varphi[:, None, None] * YY, axis=-1).transpose(3, 0, 1, 2)
    eps_act = float(np.quantile(np.linalg.eigvalsh(U).min(axis=-1), 0.1))
    cases = ((Y, dict(), dict()),
             (Y, dict(normalization=False, max_iter=3), dict(normalization=False, max_iter=3)),
             (Y, dict(flooring_fn=functools.partial(add_flooring, eps=1e-3)),
              dict(flooring=("add", 1e-3))),
             (Ymix, dict(flooring_fn=functools.partial(max_flooring, eps=eps_act), max_iter=6),
              dict(flooring=("max", eps_act), max_iter=6)),
             (Ymix, dict(), dict()),
             (Y, dict(flooring_fn=None), dict(flooring=lambda x: x)))
    for Yin, kw, okw in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = update_by_ipa(Yin, varphi, **kw)
            ref = oracle_ipa(Yin, varphi, **okw)
        # (the unnormalised problem stopped after 3 Newton steps amplifies rounding: 1e-9 at 5
        #  sources; everything else agrees to 1e-12)
        assert rel_err(a, ref) < 1e-7, kw


@pytest.mark.parametrize("N", [3, 6, 8, 9])
def test_ipa_sweep_with_singular_bins_in_the_wave(N):
    """Bins whose linear term vanishes exactly (channels with disjoint time supports: diagonal
    statistics, v = 0, the reference's singular branch, ssspy/linalg/lqpqm.py:84-93) sit at lane 0
    and elsewhere in waves whose other bins vote on the Newton step count: the other bins must come
    out as the oracle's (the mixture-wide vote is theirs alone), the singular ones -- defined by the
    reference up to the phases LAPACK leaves in its eigenvectors -- finite."""
    from oracle.ipa import update_by_ipa as oracle_ipa
    from ssspy_amd.bss._update_spatial_model import update_by_ipa

    rng = np.random.default_rng(900 + N)
    F, T = 130, 12 * N
    Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    varphi = 1.0 / (rng.random((N, F, T)) + 0.05)
    singular = [0, 5, 64, 65, 129]
    seg = T // N
    for f in singular:
        for n in range(N):
            mask = np.zeros(T)
            mask[n * seg:(n + 1) * seg] = 1.0
            Y[n, f] *= mask
    regular = np.setdiff1d(np.arange(F), singular)
    for kw in (dict(), dict(max_iter=30)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = update_by_ipa(Y, varphi, **kw)
            ref = oracle_ipa(Y, varphi, **kw)
        assert np.isfinite(a).all()
        assert rel_err(a[:, regular], ref[:, regular]) < 1e-11, kw


@pytest.mark.parametrize("kind", ["t", "ggd"])
def test_heavy_tailed_ip2_above_eight_sources_against_oracle(kind):
    """TILRMA / GGDILRMA with IP2 at 9 sources: the covariance weights of the run-time-N path need
    |W x|^2, which the standalone covariance entry now forms itself (a wide-source fuzz draw found it
    refusing: only the fused IP1 iteration had the separated power at hand).  Partitioning stays at
    8 sources and says so."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GGDILRMA, TILRMA, GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    rng = np.random.default_rng(91)
    N, F, T, K = 9, 33, 47, 5
    X = nmf_mixture(17, N, F, T)
    basis = rng.random((N, F, K)) + 0.05
    act = rng.random((N, K, T)) + 0.05
    kw = dict(n_basis=K, spatial_algorithm="IP2", source_algorithm="MM")
    m = TILRMA(dof=4.0, **kw) if kind == "t" else GGDILRMA(beta=1.0, **kw)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    ref = GaussILRMAOracle(model=("t", 4.0) if kind == "t" else ("ggd", 1.0), **kw)
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act)
    assert rel_err(Y, Yr) < 1e-6
    np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-8)
    with pytest.raises(NotImplementedError, match="partitioning takes up to 8 sources"):
        GaussILRMA(n_basis=K, partitioning=True)(X, n_iter=1)


@pytest.mark.parametrize("N", [9, 13, 16])
def test_ipa_above_eight_sources_against_oracle(N):
    """Round 6: IPA with the source count at run time (ipa_rt.hip, 9..16 sources: the reference has no
    limit, ssspy/bss/ilrma.py:180) against the oracle: the three floors -- the max floor made to act
    on a tenth of the statistics (eigen route) --, both normalisations, several Newton step counts,
    and a ragged last block of bins (70 = 64 + 6)."""
    from oracle.ipa import update_by_ipa as oracle_ipa
    from ssspy_amd.bss._update_spatial_model import update_by_ipa
    from ssspy_amd.special.flooring import add_flooring, max_flooring

    rng = np.random.default_rng(170 + N)
    F, T = (70 if N == 9 else 7), 12 * N
    Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    varphi = 1.0 / (rng.random((N, F, T)) + 0.05)
    A = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
    Ymix = 0.1 * np.einsum("fnm,mft->nft", A, Y)
    YY = Ymix[:, None] * Ymix[None, :].conj()
    U = np.mean(varphi[:, None, None] * YY, axis=-1).transpose(3, 0, 1, 2)
    eps_act = float(np.quantile(np.linalg.eigvalsh(U).min(axis=-1), 0.1))
    cases = ((Y, dict(), dict()),
             (Y, dict(normalization=False, max_iter=3), dict(normalization=False, max_iter=3)),
             (Y, dict(flooring_fn=functools.partial(add_flooring, eps=1e-3)),
              dict(flooring=("add", 1e-3))),
             (Ymix, dict(flooring_fn=functools.partial(max_flooring, eps=eps_act), max_iter=6),
              dict(flooring=("max", eps_act), max_iter=6)),
             (Y, dict(flooring_fn=None), dict(flooring=lambda x: x)))
    for Yin, kw, okw in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = update_by_ipa(Yin, varphi, **kw)
            ref = oracle_ipa(Yin, varphi, **okw)
        assert rel_err(a, ref) < 1e-7, kw


@pytest.mark.parametrize("algo,N,B", [("ISS2", 4, 1), ("IPA", 3, 1), ("ISS2", 5, 3), ("IPA", 4, 2),
                                       ("ISS2", 10, 1), ("ISS1", 4, 1), ("ISS1", 3, 3), ("ISS1", 2, 1)])
def test_ilrma_folded_power_normalization_equals_three_pass_form(algo, N, B, monkeypatch):
    """Round 5: ISS / ISS2 / IPA iterations of GaussILRMA (i) reading the mixture through the filters
    their updates imply, Y formed on demand (the default), (ii) on Y with the power normalisation
    folded into the update matrix (psi from g^H C g, C <- G C G^H, tracked log-determinant;
    _routes "implied_filter") against (iii) the literal update -> mean |y|^2 -> y / psi passes
    (_routes "folded_norm"), 12 iterations, with the loss recorded, and against the oracle."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    F, T, K = 19, 64, 3
    X = np.stack([nmf_mixture(900 + b, N, F, T) for b in range(B)])
    rng = np.random.default_rng(8)
    kw = dict(basis=rng.random((B, N, F, K)), activation=rng.random((B, N, K, T)))
    if B == 1:
        X, kw = X[0], {k: v[0] for k, v in kw.items()}

    def run():
        m = GaussILRMA(n_basis=K, spatial_algorithm=algo)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Y = m(X, n_iter=12, **{k: v.copy() for k, v in kw.items()})
        return m, Y

    if algo == "ISS1":  # (small shapes keep the fused sweep by default)
        monkeypatch.setitem(_routes.VALUES, "iss1_statistics", True)
    m0, Y0 = run()
    assert (m0._implied_filter() is not None and getattr(m0, "_ycov", None) is None) or N > 4
    monkeypatch.setitem(_routes.VALUES, "implied_filter", False)
    m1, Y1 = run()
    assert getattr(m1, "_ycov", None) is not None
    monkeypatch.setitem(_routes.VALUES, "folded_norm", False)
    m2, Y2 = run()
    monkeypatch.setitem(_routes.VALUES, "folded_norm", True)
    monkeypatch.setitem(_routes.VALUES, "implied_filter", True)
    assert getattr(m2, "_ycov", None) is None
    err = rel_err  # (after projection back: no pairwise phase ambiguity left)
    for m, Y in ((m0, Y0), (m1, Y1)):
        assert err(Y, Y2) < 1e-9
        np.testing.assert_allclose(m.loss, m2.loss, rtol=1e-9)
        assert rel_err(m.basis, m2.basis) < 1e-9
        assert rel_err(m.activation, m2.activation) < 1e-9
    if N <= 8:
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Yr = ref.run(X if B == 1 else X[0], n_iter=12,
                         **{k: (v if B == 1 else v[0]).copy() for k, v in kw.items()})
        for m, Y in ((m0, Y0), (m1, Y1)):
            loss = np.asarray(m.loss)
            np.testing.assert_allclose(loss if B == 1 else loss[:, 0], ref.loss, rtol=LOSS_RTOL)
            assert err(Y if B == 1 else Y[0], Yr) < 1e-7


@pytest.mark.parametrize("family,algo,N,F,T,seed", [
    ("ilrma", "ISS2", 3, 33, 8, 0), ("ilrma", "ISS2", 3, 33, 8, 3), ("ilrma", "ISS2", 4, 31, 11, 6),
    ("ilrma", "IPA", 4, 31, 11, 4), ("iva", "ISS2", 4, 31, 64, "ill"), ("iva", "IPA", 3, 24, 48, "ill")])
def test_implied_filter_route_is_left_past_its_rounding_bound(family, algo, N, F, T, seed, monkeypatch):
    """Round 6: the implied-filter route forms its statistics as W U W^H, which can round by
    eps * kappa where the reference's sum over the samples rounds by eps; every launch measures the
    power-weighted kappa_rms and the separators return to the on-Y iteration past 1e6
    (_device_state.py: _amp_exceeded) -- round 5 had a fence of 16 frames per source fitted to a
    fuzz draw instead.  Badly conditioned draws (3 sources on 8 frames, 4 on 11: through the filters
    alone they end 1e-6 / 6e-9 from the oracle after 12 / 8 iterations, profiles/r06_implied_guard.txt)
    and AuxIVA on a mixture with two channels equal to 1e-3 (the advisor's boundary case): the
    default run must (i) leave the route where the bound is passed, (ii) match the oracle like the
    on-Y route does; with the guard off the measured kappa_rms is above the limit."""
    import torch

    from oracle.ilrma import GaussILRMAOracle
    from oracle.iva import AuxIVAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    K, n_iter = 8, 12
    if seed == "ill":
        X = nmf_mixture(7100, N, F, T)
        X[1] = X[0] + 1e-3 * X[1]
        rng = np.random.default_rng(0)
    else:
        X = nmf_mixture(7000 + seed, N, F, T)
        rng = np.random.default_rng(seed)
    kw = dict(basis=rng.random((N, F, K)), activation=rng.random((N, K, T))) if family == "ilrma" else {}

    def run(limit=None, on_y=False):
        m = (GaussILRMA(n_basis=K, spatial_algorithm=algo) if family == "ilrma"
             else AuxLaplaceIVA(spatial_algorithm=algo))
        if limit is not None:
            m._implied_amp_limit = limit
        if on_y:
            monkeypatch.setitem(_routes.VALUES, "implied_filter", False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Y = m(X, n_iter=n_iter, **{k: v.copy() for k, v in kw.items()})
        if on_y:
            monkeypatch.setitem(_routes.VALUES, "implied_filter", True)
        torch.cuda.synchronize()
        return m, Y

    ref = (GaussILRMAOracle(n_basis=K, spatial_algorithm=algo) if family == "ilrma"
           else AuxIVAOracle(spatial_algorithm=algo, contrast="laplace"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Yr = ref.run(X, n_iter=n_iter, **{k: v.copy() for k, v in kw.items()})
    m_y, Y_y = run(on_y=True)
    m_g, Y_g = run()
    m_u, Y_u = run(limit=float("inf"))
    e_y, e_g = rel_err(Y_y, Yr), rel_err(Y_g, Yr)
    limit, kappa = type(m_u)._implied_amp_limit, m_u._amp_kappa_rms()
    print("kappa_rms {:.1e}; from the oracle: on Y {:.1e}, default {:.1e} ({} iterations through the "
          "filters), guard off {:.1e}".format(kappa, e_y, e_g, m_g._implied_iterations(),
                                              rel_err(Y_u, Yr)))
    assert m_u._implied is not None
    if T <= 8:
        assert kappa > limit
    if kappa > limit:
        assert m_g._implied is None and 0 < m_g._implied_iterations() < n_iter
    assert e_g < max(10 * e_y, 1e-11), (e_g, e_y, rel_err(Y_u, Yr))
    np.testing.assert_allclose(m_g.loss, ref.loss, rtol=max(1e3 * e_y, LOSS_RTOL))


@pytest.mark.parametrize("model,algo", [(("gauss", None), "ISS1"), (("gauss", None), "IPA"),
                                        (("gauss", None), "ISS2")])
def test_ilrma_implied_filter_iterations_mixed_with_single_steps(model, algo, monkeypatch):
    """The output state read through the implied filters stays coherent with everything that looks
    at or rewrites ``output`` in between: a callback reading it every iteration, single steps
    (update_spatial_model + normalize, which rewrite Y and retire the filters), projection back --
    each against the same sequence with the implied
    filters switched off (_routes "implied_filter")."""
    from ssspy_amd.utils.dataset import nmf_mixture

    cls, extra = _ilrma_class(model), {}
    N, F, T, K = 3, 17, 50, 2
    X = nmf_mixture(77, N, F, T)
    rng = np.random.default_rng(5)
    kw = dict(basis=rng.random((N, F, K)), activation=rng.random((N, K, T)))

    if algo == "ISS1":  # (small shapes keep the fused sweep by default)
        monkeypatch.setitem(_routes.VALUES, "iss1_statistics", True)

    def run():
        seen = []
        m = cls(n_basis=K, spatial_algorithm=algo, record_loss=True,
                callbacks=lambda mm: seen.append(np.array(mm.output)), **extra)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m(X, n_iter=3, **{k: v.copy() for k, v in kw.items()})
            lazy_between = m._implied_filter() is not None
            m.update_once()
            m.update_source_model()
            m.update_spatial_model()      # rewrites Y: the filters are retired
            m.normalize()
            m.update_once()
            loss_mid = m.compute_loss()
            for _ in range(3):
                m.update_once()
            m.restore_scale()
            Y = np.array(m.output)
        return m, Y, seen, loss_mid, lazy_between

    m0, Y0, seen0, l0, lazy0 = run()
    monkeypatch.setitem(_routes.VALUES, "implied_filter", False)
    m1, Y1, seen1, l1, lazy1 = run()
    monkeypatch.setitem(_routes.VALUES, "implied_filter", True)
    assert lazy0 and not lazy1
    assert len(seen0) == len(seen1) > 0
    for a, b in zip(seen0, seen1):
        assert rel_err(a, b) < 1e-9
    assert rel_err(Y0, Y1) < 1e-9
    np.testing.assert_allclose(m0.loss, m1.loss, rtol=1e-9)
    np.testing.assert_allclose(l0, l1, rtol=1e-9)
    assert rel_err(m0.basis, m1.basis) < 1e-9


@pytest.mark.parametrize("contrast,algo,N,B", [("laplace", "ISS2", 4, 1), ("laplace", "IPA", 3, 2),
                                               ("gauss", "ISS2", 3, 1), ("gauss", "IPA", 4, 3),
                                               ("laplace", "ISS2", 2, 1)])
def test_auxiva_implied_filter_iterations_equal_the_literal_form(contrast, algo, N, B, monkeypatch):
    """Round 5: ISS2 / IPA iterations of AuxIVA reading the mixture through the filters their updates
    imply (frame powers |W x|^2, statistics W U W^H, W <- G W; Y formed on read) against the literal
    passes over Y (_routes "implied_filter") and the oracle: outputs seen by a callback every
    iteration, the loss list, the result after projection back; then single steps that rewrite Y."""
    from oracle.iva import AuxIVAOracle
    from ssspy_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    F, T = 21, 70
    X = np.stack([nmf_mixture(300 + b, N, F, T) for b in range(B)])
    if B == 1:
        X = X[0]
    cls = AuxLaplaceIVA if contrast == "laplace" else AuxGaussIVA

    def run(with_callback):
        seen = []
        kw = {"callbacks": (lambda mm: seen.append(np.array(mm.output)))} if with_callback else {}
        m = cls(spatial_algorithm=algo, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Y = np.array(m(X, n_iter=6))
            lazy = m._implied_filter() is not None
            m.update_once()
            loss_mid = m.compute_loss()
            Ymid = np.array(m.output)
        return m, Y, seen, lazy, loss_mid, Ymid

    a = run(False)
    b = run(True)
    monkeypatch.setitem(_routes.VALUES, "implied_filter", False)
    c = run(True)
    monkeypatch.setitem(_routes.VALUES, "implied_filter", True)
    assert a[3] and b[3] and not c[3]
    # Two sources with the default pairs (0, 1), (1, 0): the second problem of every iteration is the
    # pair the first one has just diagonalised, its off-diagonal entry is rounding noise and the
    # phase np.linalg.eigh -- hence the reference, and csrc/eigh2.hpp since round 6 -- gives the
    # eigenvectors is derived from it: the rows of an output that has NOT been through scale
    # restoration are then defined in modulus only (benchmarks/tools/pair_phase_check.py).
    unrestored = (lambda y: np.abs(y)) if (algo == "ISS2" and N == 2) else (lambda y: y)
    for r in (a, b):
        assert rel_err(r[1], c[1]) < 1e-9
        np.testing.assert_allclose(r[0].loss, c[0].loss, rtol=1e-9)
        np.testing.assert_allclose(r[4], c[4], rtol=1e-9)
        assert rel_err(unrestored(r[5]), unrestored(c[5])) < 1e-9
    assert len(b[2]) == len(c[2]) == 7
    for u, v in zip(b[2], c[2]):
        assert rel_err(unrestored(u), unrestored(v)) < 1e-9
    ref = AuxIVAOracle(spatial_algorithm=algo, contrast=contrast)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Yr = ref.run(X if B == 1 else X[0], n_iter=6)
    loss = np.asarray(a[0].loss)
    np.testing.assert_allclose(loss if B == 1 else loss[:, 0], ref.loss, rtol=LOSS_RTOL)
    assert rel_err(a[1] if B == 1 else a[1][0], Yr) < 1e-7


def test_ipa_eight_sources_against_oracle():
    from oracle.ipa import update_by_ipa as oracle_ipa
    from ssspy_amd.bss._update_spatial_model import update_by_ipa

    rng = np.random.default_rng(208)
    N, F, T = 8, 5, 60
    Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    varphi = 1 / (rng.random((N, F, T)) + 0.1)
    assert rel_err(update_by_ipa(Y, varphi, max_iter=2), oracle_ipa(Y, varphi, max_iter=2)) < 1e-10


def test_ip1_singular_raises_linalgerror():
    from ssspy_amd.bss._update_spatial_model import update_by_ip1

    W = np.zeros((3, 2, 2), dtype=complex)
    U = np.tile(np.eye(2, dtype=complex), (3, 2, 1, 1))
    with pytest.raises(np.linalg.LinAlgError):
        update_by_ip1(W, U)


# ------------------------------------------------------------------------------- GaussILRMA
def _ilrma_class(model):
    """(class, extra ctor kwargs) for a source model given as ("gauss"|"t"|"ggd", param)."""
    from ssspy_amd.bss.ilrma import GGDILRMA, TILRMA, GaussILRMA

    if model[0] == "t":
        return functools.partial(TILRMA, dof=model[1])
    if model[0] == "ggd":
        return functools.partial(GGDILRMA, beta=model[1])
    return GaussILRMA


@pytest.mark.parametrize("case", ILRMA_CASES)
def test_gauss_ilrma_against_golden(case):
    g = load_golden(case)
    model = (str(g["meta_model"]), float(g["meta_model_param"])) if "meta_model" in g \
        else ("gauss", None)
    snap = Snap(["demix_filter", "output", "basis", "activation", "latent"])
    partitioning = bool(g["meta_partitioning"]) if "meta_partitioning" in g else False
    m = _ilrma_class(model)(
        n_basis=int(g["meta_n_basis"]), spatial_algorithm=str(g["meta_algo"]),
        partitioning=partitioning,
        domain=float(g["meta_domain"]), flooring_fn=_flooring_fn(g), callbacks=snap,
        normalization=_option(g["meta_normalization"]),
        scale_restoration=_option(g["meta_scale_restoration"]),
        source_algorithm=str(g["meta_source_algorithm"]) if "meta_source_algorithm" in g else "MM",
        **({"newton_iter": int(g["meta_newton_iter"])} if str(g["meta_algo"]) == "IPA" else {}),
    )
    b0, a0 = g["basis0"].copy(), g["activation0"].copy()
    extra = {"latent": g["latent0"].copy()} if partitioning else {}
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]), basis=b0, activation=a0, **extra)
    assert np.array_equal(b0, g["basis0"]) and np.array_equal(a0, g["activation0"])  # not mutated
    if partitioning:
        assert np.array_equal(extra["latent"], g["latent0"])
        assert rel_err(m.latent, g["final_latent"]) < TOL
        assert m.basis.shape == g["final_basis"].shape
    _compare_snapshots(g, snap)
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    assert all(type(v) is float for v in m.loss)
    assert Y.shape == g["X"].shape and Y.dtype == np.complex128
    assert rel_err(Y, g["final_output"]) < TOL
    if "final_demix_filter" in g:
        assert rel_err(m.demix_filter, g["final_demix_filter"]) < TOL
    else:
        assert m.demix_filter is None
    assert rel_err(m.basis, g["final_basis"]) < TOL
    assert rel_err(m.activation, g["final_activation"]) < TOL


def test_gauss_ilrma_step_methods_match_fused_update():
    """update_once() as one C-ABI call == the public per-step methods called one by one."""
    from ssspy_amd.bss.ilrma import GaussILRMA

    g = load_golden("gilrma_ip1_n4")

    class Stepwise(GaussILRMA):
        def normalize(self, flooring_fn="self"):  # overriding forces the step-by-step path
            super().normalize(flooring_fn=flooring_fn)

    outs = []
    for cls in (GaussILRMA, Stepwise):
        m = cls(n_basis=int(g["meta_n_basis"]))
        outs.append(m(g["X"], n_iter=4, basis=g["basis0"], activation=g["activation0"]))
    assert rel_err(outs[1], outs[0]) < 1e-12


def test_gauss_ilrma_iss_power_reuse_is_invalidated_by_output_writes():
    """The ISS sweep leaves the frame powers of its result for the normalisation that follows; a
    write to ``output`` in between (host assignment here) must make normalize() measure Y again."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 3, 20, 40, 4
    X = nmf_mixture(77, N, F, T)
    rng = np.random.default_rng(3)
    basis, act = rng.random((N, F, K)), rng.random((N, K, T))
    m = GaussILRMA(n_basis=K, spatial_algorithm="ISS")
    m(X, n_iter=1, basis=basis, activation=act)
    m.update_source_model()
    m.update_spatial_model()
    Y = m.output
    scale = np.array([2.0, 0.5, 3.0])[:, None, None]
    m.output = Y * scale  # power changes by scale^2; the cached frame powers describe the old Y
    basis_before = m.basis
    m.normalize()
    Yn = m.output
    np.testing.assert_allclose(np.mean(np.abs(Yn) ** 2, axis=(1, 2)), 1.0, rtol=1e-12)
    psi2 = np.mean(np.abs(Y * scale) ** 2, axis=(1, 2))
    assert rel_err(m.basis, basis_before / psi2[:, None, None]) < 1e-12


@pytest.mark.parametrize("model", [("t", 3.0), ("ggd", 1.2)])
@pytest.mark.parametrize("N,K,algo,domain", [(2, 2, "ISS2", 2), (4, 16, "IP", 2), (5, 20, "IP2", 1),
                                             (8, 3, "ISS", 2), (4, 8, "IP1", 1.5)])
def test_heavy_tailed_ilrma_sweep_against_oracle(model, N, K, algo, domain):
    """TILRMA / GGDILRMA over source counts, n_basis (incl. > 16), domains and all spatial updates."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.utils.dataset import nmf_mixture

    F, T = 19, 41
    X = nmf_mixture(11 + N, N, F, T)
    basis = np.random.default_rng(3).random((N, F, K))
    act = np.random.default_rng(4).random((N, K, T))
    ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, domain=domain, model=model,
                           scale_restoration=False)
    Yr = ref.run(X, n_iter=4, basis=basis, activation=act)
    m = _ilrma_class(model)(n_basis=K, spatial_algorithm=algo, domain=domain,
                            scale_restoration=False)
    Y = m(X, n_iter=4, basis=basis, activation=act)
    err = rel_err_up_to_phase(Y, Yr, "output") if algo in ("IP2", "ISS2") else rel_err(Y, Yr)
    assert err < TOL
    assert rel_err(m.basis, ref.basis) < TOL and rel_err(m.activation, ref.activation) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)


@pytest.mark.parametrize("algo", ["IP", "ISS", "IP2"])
@pytest.mark.parametrize("N,K,F,T,domain", [(4, 16, 70, 130, 1), (3, 5, 33, 47, 1), (2, 16, 20, 64, 1),
                                            (4, 7, 33, 48, 1.5), (4, 16, 70, 130, 2),
                                            (3, 16, 70, 130, 0.6), (4, 12, 40, 100, 1.7)])
def test_gauss_ilrma_domain_sweep_against_oracle(algo, N, K, F, T, domain):
    """Gauss-ILRMA off the power domain on the tuned kernels: domain 1 with R^3 / R^2 instead of
    powers and a cube-root update, any other domain in (0, 2) with the powers as exp2(e log2 R)
    (round 3; the generic kernels before); several frame tiles, ragged edges, split and unsplit
    work items."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(40 + N + F, N, F, T)
    rng = np.random.default_rng(N * K)
    basis, act = rng.random((N, F, K)), rng.random((N, K, T))
    ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, domain=domain)
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act)
    m = GaussILRMA(n_basis=K, spatial_algorithm=algo, domain=domain)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    assert rel_err(Y, Yr) < TOL
    assert rel_err(m.basis, ref.basis) < TOL and rel_err(m.activation, ref.activation) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)


@pytest.mark.parametrize("beta", [0.5, 1.0, 1.9])
@pytest.mark.parametrize("algo", ["IP", "ISS"])
def test_ggd_ilrma_tuned_kernels_against_oracle(beta, algo):
    """GGD-ILRMA at domain 2 on the tuned kernels: shape parameters across (0, 2), 4 sources,
    n_basis 16, several frame tiles, and a batch large enough for unsplit work items."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GGDILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, K, F, T = 4, 16, 70, 130
    X = nmf_mixture(91, N, F, T)
    rng = np.random.default_rng(17)
    basis, act = rng.random((N, F, K)), rng.random((N, K, T))
    ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, model=("ggd", beta))
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act)
    m = GGDILRMA(n_basis=K, beta=beta, spatial_algorithm=algo)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    assert rel_err(Y, Yr) < TOL
    assert rel_err(m.basis, ref.basis) < TOL and rel_err(m.activation, ref.activation) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)
    nb = 260  # 520 (mixture, 64-bin group) items: 512 unsplit + 8 split
    Xb = np.stack([X] * nb)
    mb = GGDILRMA(n_basis=K, beta=beta, spatial_algorithm=algo)
    Yb = mb(Xb, n_iter=3, basis=np.stack([basis] * nb), activation=np.stack([act] * nb))
    assert rel_err(Yb[0], Y) < 1e-12 and rel_err(Yb[-1], Y) < 1e-12


def test_heavy_tailed_ilrma_batch_and_stepwise():
    """Batched TILRMA == per-mixture runs; fused update_once == the step methods."""
    from ssspy_amd.bss.ilrma import TILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 3, 33, 64, 4
    Xb = np.stack([nmf_mixture(s, N, F, T) for s in (1, 2, 3)])
    basis = np.random.default_rng(5).random((3, N, F, K))
    act = np.random.default_rng(6).random((3, N, K, T))
    Yb = TILRMA(n_basis=K, dof=4.0)(Xb, n_iter=3, basis=basis, activation=act)

    class Stepwise(TILRMA):
        def normalize(self, flooring_fn="self"):
            super().normalize(flooring_fn=flooring_fn)

    for b in range(3):
        for cls in (TILRMA, Stepwise):
            Y = cls(n_basis=K, dof=4.0)(Xb[b], n_iter=3, basis=basis[b], activation=act[b])
            assert rel_err(Y, Yb[b]) < 1e-12


@pytest.mark.parametrize("model,algo,src", [(("gauss", None), "ISS2", "MM"), (("t", 3.0), "IP", "ME"),
                                            (("ggd", 1.4), "IP2", "MM")])
def test_partitioned_ilrma_batch_against_oracle(model, algo, src):
    """partitioning=True with n_basis > 16 and 5 sources, batched == per-mixture oracle runs."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 5, 21, 46, 20
    Xb = np.stack([nmf_mixture(s, N, F, T) for s in (41, 42)])
    basis = np.random.default_rng(9).random((2, F, K))
    act = np.random.default_rng(10).random((2, K, T))
    lat = np.random.default_rng(11).random((2, N, K))
    lat = lat / lat.sum(axis=1, keepdims=True)
    m = _ilrma_class(model)(n_basis=K, spatial_algorithm=algo, source_algorithm=src,
                            partitioning=True, scale_restoration=False)
    Yb = m(Xb, n_iter=3, basis=basis, activation=act, latent=lat)
    assert m.latent.shape == (2, N, K) and m.basis.shape == (2, F, K)
    for b in range(2):
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, model=model, source_algorithm=src,
                               partitioning=True, scale_restoration=False)
        Yr = ref.run(Xb[b], n_iter=3, basis=basis[b], activation=act[b], latent=lat[b])
        err = rel_err_up_to_phase(Yb[b], Yr, "output") if algo in ("IP2", "ISS2") else rel_err(Yb[b], Yr)
        assert err < TOL
        assert rel_err(m.latent[b], ref.latent) < TOL and rel_err(m.basis[b], ref.basis) < TOL
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref.loss, rtol=LOSS_RTOL)


def test_partitioned_ilrma_many_bases_against_oracle():
    """partitioning=True with n_basis = 300 (the latent variables of all sources sit in LDS of one
    workgroup: sized for n_basis <= 1024) and n_basis beyond the limit refused loudly."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 3, 12, 30, 300
    X = nmf_mixture(44, N, F, T)
    rng = np.random.default_rng(12)
    basis, act = rng.random((F, K)), rng.random((K, T))
    lat = rng.random((N, K))
    lat = lat / lat.sum(axis=0, keepdims=True)
    m = GaussILRMA(n_basis=K, partitioning=True)
    Y = m(X, n_iter=3, basis=basis, activation=act, latent=lat)
    ref = GaussILRMAOracle(n_basis=K, partitioning=True)
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act, latent=lat)
    assert rel_err(Y, Yr) < TOL and rel_err(m.latent, ref.latent) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)
    with pytest.raises((NotImplementedError, ValueError)):
        GaussILRMA(n_basis=1025, partitioning=True)(X, n_iter=1)


@pytest.mark.parametrize("cls", ["ilrma_ip", "ilrma_iss", "fast", "gauss"])
def test_n_basis_above_1024_against_oracle(cls):
    """The reference puts no bound on n_basis (ssspy/bss/ilrma.py:201-270, mnmf.py:1112-1153); the
    entry points refused more than 1024 until round 6 although the dense-product kernels walk any
    n_basis (round-5 verdict, missing item 5).  1500 bases, tiny spectrograms."""
    from oracle.gmnmf import GaussMNMFOracle
    from oracle.ilrma import GaussILRMAOracle
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 3, 33, 17, 1500
    X = nmf_mixture(5, N, F, T)
    kw = dict(basis=np.random.default_rng(1).random((N, F, K)),
              activation=np.random.default_rng(2).random((N, K, T)))
    if cls.startswith("ilrma"):
        algo = "IP" if cls == "ilrma_ip" else "ISS"
        ref, m = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo), GaussILRMA(n_basis=K, spatial_algorithm=algo)
    elif cls == "fast":
        kw["spatial"] = np.random.default_rng(3).random((F, N, N)) + 0.05
        ref, m = FastGaussMNMFOracle(n_basis=K), FastGaussMNMF(n_basis=K)
    else:
        ref = GaussMNMFOracle(n_basis=K, rng=np.random.default_rng(1))
        m = GaussMNMF(n_basis=K, rng=np.random.default_rng(1))
    Yr = ref.run(X, n_iter=3, **{k: v.copy() for k, v in kw.items()})
    Y = m(X, n_iter=3, **kw)
    assert rel_err(Y, Yr) < 1e-8
    np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-9)
    assert rel_err(m.basis, ref.basis) < 1e-8 and rel_err(m.activation, ref.activation) < 1e-8


def test_heavy_tailed_ilrma_constructor_contract():
    from ssspy_amd.bss.ilrma import GGDILRMA, TILRMA

    with pytest.raises(ValueError):
        TILRMA(n_basis=2, dof=3, spatial_algorithm="IPA")
    with pytest.raises(ValueError):
        GGDILRMA(n_basis=2, beta=1.0, spatial_algorithm="IPA")
    with pytest.raises(AssertionError):
        GGDILRMA(n_basis=2, beta=2.0)
    with pytest.raises(AssertionError):
        GGDILRMA(n_basis=2, beta=1.0, source_algorithm="ME")
    assert "dof=3" in repr(TILRMA(n_basis=2, dof=3)) and repr(TILRMA(2, 3)).startswith("TILRMA(")


@pytest.mark.parametrize("K", [1, 7, 16, 17, 24, 32, 33, 40, 48, 64, 65, 100, 300])
def test_gauss_ilrma_n_basis_sweep_against_oracle(K):
    """n_basis off the 16-wide MFMA tile (1, 7), on it (16), on the two- and four-k-tile variants of
    the tuned kernels (17..32, 33..64) and beyond them (65, 100: generic kernels)."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T = 3, 21, 45
    X = nmf_mixture(7, N, F, T)
    basis = np.random.default_rng(1).random((N, F, K))
    act = np.random.default_rng(2).random((N, K, T))
    ref = GaussILRMAOracle(n_basis=K)
    Yr = ref.run(X, n_iter=5, basis=basis, activation=act)
    m = GaussILRMA(n_basis=K)
    Y = m(X, n_iter=5, basis=basis, activation=act)
    assert rel_err(Y, Yr) < TOL
    assert rel_err(m.basis, ref.basis) < TOL and rel_err(m.activation, ref.activation) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)


@pytest.mark.parametrize("K,algo", [(40, "IP"), (64, "IP"), (48, "ISS")])
def test_gauss_ilrma_wide_basis_batch_equals_single_and_oracle(K, algo):
    """33 <= n_basis <= 64 on a batch large enough for unsplit and split work items (four k-tile
    items per bin group): elements equal their single-mixture runs, element 0 the oracle."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, N, F, T = 70, 4, 70, 50
    X = np.stack([nmf_mixture(400 + b, N, F, T) for b in range(B)])
    rng = np.random.default_rng(K)
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    m = GaussILRMA(n_basis=K, spatial_algorithm=algo)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    for b in (0, 33, 69):
        s1 = GaussILRMA(n_basis=K, spatial_algorithm=algo)
        Yb = s1(X[b], n_iter=3, basis=basis[b], activation=act[b])
        assert rel_err(Y[b], Yb) < 1e-11, b
        assert rel_err(m.basis[b], s1.basis) < 1e-11 and rel_err(m.activation[b], s1.activation) < 1e-11
    ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
    Yr = ref.run(X[0], n_iter=3, basis=basis[0], activation=act[0])
    assert rel_err(Y[0], Yr) < TOL


@pytest.mark.parametrize("N,algo", [(2, "IP"), (5, "IP"), (6, "ISS"), (8, "IP")])
def test_gauss_ilrma_source_count_sweep_against_oracle(N, algo):
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    F, T, K = 18, 40, 4
    X = nmf_mixture(11 + N, N, F, T)
    basis = np.random.default_rng(1).random((N, F, K))
    act = np.random.default_rng(2).random((N, K, T))
    ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
    Yr = ref.run(X, n_iter=4, basis=basis, activation=act)
    m = GaussILRMA(n_basis=K, spatial_algorithm=algo)
    Y = m(X, n_iter=4, basis=basis, activation=act)
    assert rel_err(Y, Yr) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)


def test_gauss_ilrma_batch_equals_single():
    """A 4-D batch of independent mixtures gives, per mixture, what the 3-D call gives."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K, B = 4, 33, 48, 6, 3
    Xs = np.stack([nmf_mixture(100 + b, N, F, T) for b in range(B)])
    basis = np.random.default_rng(1).random((B, N, F, K))
    act = np.random.default_rng(2).random((B, N, K, T))
    mb = GaussILRMA(n_basis=K)
    Yb = mb(Xs, n_iter=5, basis=basis, activation=act)
    assert Yb.shape == Xs.shape and np.asarray(mb.loss).shape == (6, B)
    for b in range(B):
        m = GaussILRMA(n_basis=K)
        Y = m(Xs[b], n_iter=5, basis=basis[b], activation=act[b])
        assert rel_err(Yb[b], Y) < 1e-12
        np.testing.assert_allclose(np.asarray(mb.loss)[:, b], m.loss, rtol=1e-12)


def test_gauss_ilrma_config2_full_size_against_oracle():
    """BASELINE.json configs[1] shape (N=4, F=1025, T=512, K=16), 2 iterations vs the oracle."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 4, 1025, 512, 16
    X = nmf_mixture(1000, N, F, T)
    basis = np.random.default_rng(1001).random((N, F, K))
    act = np.random.default_rng(1002).random((N, K, T))
    ref = GaussILRMAOracle(n_basis=K)
    Yr = ref.run(X, n_iter=2, basis=basis, activation=act)
    m = GaussILRMA(n_basis=K)
    Y = m(X, n_iter=2, basis=basis, activation=act)
    assert rel_err(m.demix_filter, ref.demix_filter) < TOL
    assert rel_err(Y, Yr) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)


def test_gauss_ilrma_full_size_properties():
    """Size-independent properties at configs[1] size over 30 iterations: the loss is
    non-increasing (MM + IP guarantee), the power normalisation holds (mean |y_n|^2 = 1 before
    scale restoration), and projection back makes the reference-channel reconstruction exact
    (sum_n y_n = x_ref)."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 4, 1025, 512, 16
    X = nmf_mixture(1000, N, F, T)
    power = []

    def cb(m):
        Y = m.separate(m.input, m.demix_filter)
        power.append(np.mean(np.abs(Y) ** 2, axis=(1, 2)))

    m = GaussILRMA(n_basis=K, rng=np.random.default_rng(0), callbacks=cb)
    Y = m(X, n_iter=30)
    loss = np.array(m.loss)
    assert np.all(np.diff(loss) <= 1e-9 * np.abs(loss[:-1]))
    np.testing.assert_allclose(power[-1], 1.0, rtol=1e-10)
    assert rel_err(Y.sum(axis=0), X[0]) < 1e-10


# ------------------------------------------------------------------------------- AuxIVA
@pytest.mark.parametrize("case", IVA_CASES)
def test_aux_iva_against_golden(case):
    from ssspy_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA

    g = load_golden(case)
    contrast = str(g["meta_contrast"])
    snap = Snap(["demix_filter", "output"] + (["variance"] if contrast == "gauss" else []))
    cls = AuxLaplaceIVA if contrast == "laplace" else AuxGaussIVA
    m = cls(spatial_algorithm=str(g["meta_algo"]), flooring_fn=_flooring_fn(g), callbacks=snap,
            scale_restoration=_option(g["meta_scale_restoration"]))
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]))
    _compare_snapshots(g, snap)
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    assert rel_err(Y, g["final_output"]) < TOL
    if "final_demix_filter" in g:
        assert rel_err(m.demix_filter, g["final_demix_filter"]) < TOL
    else:
        assert m.demix_filter is None


@pytest.mark.parametrize("case,algo", [("kat_auxlap_ip1_config1", "IP"), ("kat_auxlap_iss1_config1", "ISS")])
def test_aux_iva_config1_kat(case, algo):
    """BASELINE.json configs[0] (N=2, F=257, T=128, 10 it): known-answer scalars of the reference."""
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import iid_mixture

    g = load_golden(case)
    N, F, T = (int(v) for v in g["meta_shape"])
    X = iid_mixture(int(g["meta_seed"]), N, F, T)
    m = AuxLaplaceIVA(spatial_algorithm=algo)
    Y = m(X, n_iter=int(g["meta_n_iter"]))
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    assert Y[0, 0, 0] == pytest.approx(complex(g["kat_y000"]), rel=1e-8)
    assert np.sum(np.abs(Y) ** 2) == pytest.approx(float(g["kat_energy"]), rel=1e-9)


def test_aux_iva_iss_config3_shape_against_oracle():
    """configs[2] channel count (N=8) at a size the oracle finishes quickly, ISS, 3 iterations."""
    from oracle.iva import AuxIVAOracle
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(5, 8, 65, 200)
    ref = AuxIVAOracle(spatial_algorithm="ISS", contrast="laplace")
    Yr = ref.run(X, n_iter=3)
    m = AuxLaplaceIVA(spatial_algorithm="ISS")
    Y = m(X, n_iter=3)
    assert rel_err(Y, Yr) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)


# ------------------------------------------------------------------------------- FastGaussMNMF
MNMF_CASES = ["fmnmf_ip1_m3", "fmnmf_ip1_m4", "fmnmf_ip1_m3_n2", "fmnmf_ip1_m2_nonorm",
              "fmnmf_ip1_m6_n3", "fmnmf_ip1_m5", "fmnmf_ip1_m8_n2"]


@pytest.mark.parametrize("case", MNMF_CASES)
def test_fast_gauss_mnmf_against_golden(case):
    from ssspy_amd.bss.mnmf import FastGaussMNMF

    g = load_golden(case)
    snap = Snap(["diagonalizer", "spatial", "basis", "activation"])
    m = FastGaussMNMF(n_basis=int(g["meta_n_basis"]), n_sources=int(g["meta_n_sources"]),
                      flooring_fn=_flooring_fn(g), callbacks=snap,
                      normalization=_option(g["meta_normalization"]))
    sp0 = g["spatial0"].copy()
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]), basis=g["basis0"], activation=g["activation0"],
          spatial=sp0)
    assert np.array_equal(sp0, g["spatial0"])
    _compare_snapshots(g, snap)
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    assert rel_err(m.diagonalizer, g["final_diagonalizer"]) < TOL
    assert rel_err(m.spatial, g["final_spatial"]) < TOL
    assert rel_err(Y, g["final_output"]) < 1e-7  # Wiener filter: eigh + solve, cond(R)-amplified


GMNMF_CASES = ["gmnmf_m2", "gmnmf_m3", "gmnmf_m4_n3", "gmnmf_m2_nonorm_add", "gmnmf_part_m3",
               "gmnmf_part_m2_n3", "gmnmf_m5", "gmnmf_m6_n3", "gmnmf_m7", "gmnmf_m8",
               # eigenvalue floor of to_psd active at most points (eps = 0.3), 10 iterations
               "gmnmf_floor_m5", "gmnmf_floor_m6_n3", "gmnmf_floor_m8"]


@pytest.mark.parametrize("case", GMNMF_CASES)
def test_gauss_mnmf_against_golden(case):
    """Full-rank MNMF against the reference: 10 iterations of per-point eigen-floors, inverses and
    a matrix geometric mean.  Held to 1e-7 (cond(R)-amplified round-off; the reference's own
    regression tolerance for this class is atol 1e-7)."""
    from ssspy_amd.bss.mnmf import GaussMNMF

    g = load_golden(case)
    snap = Snap(["spatial", "basis", "activation", "latent"])
    part = bool(g["meta_partitioning"]) if "meta_partitioning" in g else False
    m = GaussMNMF(n_basis=int(g["meta_n_basis"]), n_sources=int(g["meta_n_sources"]),
                  partitioning=part, flooring_fn=_flooring_fn(g), callbacks=snap,
                  normalization=_option(g["meta_normalization"]))
    init = dict(basis=g["basis0"], activation=g["activation0"])
    if "spatial0" in g:
        init["spatial"] = g["spatial0"].copy()
    if part:
        init["latent"] = g["latent0"].copy()
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]), **init)
    for key, value in snap.store.items():
        assert rel_err(value, g[key]) < 1e-7, key
    n_snap = sum(1 for it in (1, 2, 10) if it <= int(g["meta_n_iter"]))  # callbacks 1, 2, 10
    assert len(snap.store) == (4 if part else 3) * n_snap
    if part:
        assert rel_err(m.latent, g["final_latent"]) < 1e-7
    np.testing.assert_allclose(m.loss, g["loss"], rtol=1e-8)
    assert all(type(v) is float for v in m.loss)
    assert rel_err(m.spatial, g["final_spatial"]) < 1e-7
    assert rel_err(m.basis, g["final_basis"]) < 1e-7
    assert rel_err(Y, g["final_output"]) < 1e-7
    assert m.spatial.shape == g["final_spatial"].shape and Y.dtype == np.complex128


def test_gauss_mnmf_steps_batch_and_oracle():
    """Step methods == fused update; batched == per-mixture; larger shape against the oracle
    (K = 20 > 16, n_sources = 5 > n_channels = 4, T not a multiple of the block)."""
    from oracle.gmnmf import GaussMNMFOracle
    from ssspy_amd.bss.mnmf import GaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    M, N, F, T, K = 4, 5, 13, 150, 20
    Xb = np.stack([nmf_mixture(s, M, F, T) for s in (31, 32)])
    basis = np.random.default_rng(7).random((2, N, F, K))
    act = np.random.default_rng(8).random((2, N, K, T))

    class Stepwise(GaussMNMF):
        def normalize(self, axis1=-2, axis2=-1):
            super().normalize(axis1=axis1, axis2=axis2)

    mb = GaussMNMF(n_basis=K, n_sources=N)
    Yb = mb(Xb, n_iter=3, basis=basis, activation=act)
    assert np.asarray(mb.loss).shape == (4, 2)
    for b in range(2):
        ref = GaussMNMFOracle(n_basis=K, n_sources=N)
        Yr = ref.run(Xb[b], n_iter=3, basis=basis[b], activation=act[b])
        assert rel_err(Yb[b], Yr) < 1e-7
        assert rel_err(mb.spatial[b], ref.spatial) < 1e-7
        np.testing.assert_allclose(np.asarray(mb.loss)[:, b], ref.loss, rtol=1e-8)
        for cls in (GaussMNMF, Stepwise):
            m = cls(n_basis=K, n_sources=N)
            Y = m(Xb[b], n_iter=3, basis=basis[b], activation=act[b])
            assert rel_err(Y, Yb[b]) < 1e-11


def test_gauss_mnmf_active_eigenvalue_floor():
    """A large floor (0.3) makes to_psd clip eigenvalues at many points, which sends those lanes
    down the eigen-decomposition path instead of the Cholesky shortcut."""
    from oracle.gmnmf import GaussMNMFOracle
    from ssspy_amd.bss.mnmf import GaussMNMF
    from ssspy_amd.special.flooring import max_flooring
    from ssspy_amd.utils.dataset import nmf_mixture

    M, F, T, K = 3, 9, 70, 3
    X = 0.3 * nmf_mixture(51, M, F, T)
    basis = np.random.default_rng(12).random((M, F, K))
    act = np.random.default_rng(13).random((M, K, T))
    ref = GaussMNMFOracle(n_basis=K, flooring=("max", 0.3))
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act)
    lam = np.linalg.eigvalsh(np.sum((ref.basis @ ref.activation)[:, :, :, None, None]
                                    * ref.spatial[:, :, None], axis=0))
    assert 0.05 < np.mean(lam < 0.3) < 0.95  # the floor is really active, and not everywhere
    m = GaussMNMF(n_basis=K, flooring_fn=functools.partial(max_flooring, eps=0.3))
    Y = m(X, n_iter=3, basis=basis, activation=act)
    assert rel_err(Y, Yr) < 1e-8 and rel_err(m.spatial, ref.spatial) < 1e-8
    np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-9)


def test_fast_gauss_mnmf_step_methods_match_fused_update():
    from ssspy_amd.bss.mnmf import FastGaussMNMF

    g = load_golden("fmnmf_ip1_m4")

    class Stepwise(FastGaussMNMF):
        def update_spatial(self):
            super().update_spatial()

    outs = []
    for cls in (FastGaussMNMF, Stepwise):
        m = cls(n_basis=int(g["meta_n_basis"]))
        m(g["X"], n_iter=3, basis=g["basis0"], activation=g["activation0"], spatial=g["spatial0"])
        outs.append((m.diagonalizer, m.spatial, m.basis, m.activation))
    for a, b in zip(*outs):
        assert rel_err(b, a) < 1e-12


def test_fast_gauss_mnmf_config4_shape_against_oracle():
    """configs[3] channel/source/basis counts (N=M=4, K=8) at an oracle-sized F x T, 3 iterations."""
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    M, F, T, K = 4, 65, 96, 8
    X = nmf_mixture(21, M, F, T)
    basis = np.random.default_rng(1).random((M, F, K))
    act = np.random.default_rng(2).random((M, K, T))
    spatial = np.random.default_rng(4).random((F, M, M))
    ref = FastGaussMNMFOracle(n_basis=K)
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act, spatial=spatial.copy())
    m = FastGaussMNMF(n_basis=K)
    Y = m(X, n_iter=3, basis=basis, activation=act, spatial=spatial)
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)
    assert rel_err(m.diagonalizer, ref.diagonalizer) < TOL
    assert rel_err(Y, Yr) < 1e-7


def _fastmnmf_states(m):
    return [np.array(v) for v in (m.diagonalizer, m.spatial, m.basis, m.activation, m.output)]


@pytest.mark.parametrize("M,F,T,K", [(4, 70, 96, 8), (3, 33, 50, 5), (2, 17, 130, 16), (4, 20, 37, 3)])
def test_fast_gauss_mnmf_handover_matches_plain_path_and_oracle(M, F, T, K, monkeypatch):
    """The |Q x|^2 hand-over (spatial pass -> next basis / activation passes) against the passes
    that read x and Q, and against the oracle.  Odd T has no hand-over (8-byte aligned rows)."""
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(77, M, F, T)
    kw = dict(basis=np.random.default_rng(1).random((M, F, K)),
              activation=np.random.default_rng(2).random((M, K, T)),
              spatial=np.random.default_rng(4).random((F, M, M)))
    m1 = FastGaussMNMF(n_basis=K)
    m1(X, n_iter=4, **{k: v.copy() for k, v in kw.items()})
    assert (m1._handover is not None) == (T % 2 == 0)
    monkeypatch.setitem(_routes.VALUES, "handover", False)
    m2 = FastGaussMNMF(n_basis=K)
    m2(X, n_iter=4, **{k: v.copy() for k, v in kw.items()})
    assert m2._handover is None
    for a, b in zip(_fastmnmf_states(m1), _fastmnmf_states(m2)):
        assert rel_err(a, b) < 1e-11
    np.testing.assert_allclose(m1.loss, m2.loss, rtol=1e-11)
    ref = FastGaussMNMFOracle(n_basis=K)
    Yr = ref.run(X, n_iter=4, **{k: v.copy() for k, v in kw.items()})
    np.testing.assert_allclose(m1.loss, ref.loss, rtol=LOSS_RTOL)
    assert rel_err(m1.diagonalizer, ref.diagonalizer) < TOL
    assert rel_err(m1.output, Yr) < 1e-7


@pytest.mark.parametrize("B,M,N,F,T,K", [(1, 4, 4, 70, 96, 8), (1, 3, 3, 33, 48, 5), (1, 2, 2, 17, 16, 16),
                                         (1, 4, 4, 130, 32, 3), (3, 4, 4, 65, 160, 12), (2, 3, 2, 64, 64, 9),
                                         (5, 4, 3, 129, 80, 16), (40, 4, 4, 20, 48, 4)])
def test_fast_gauss_mnmf_lds_dma_passes_against_oracle(B, M, N, F, T, K):
    """Round 5: the covariance and spatial passes fed by LDS-DMA (k_mnmf_binmajor_glds, T % 16 == 0)
    against the oracle: single tiles, one and several mixtures (whole items and frame-split items),
    fewer sources than channels, n_basis on both sides of the k-slab variants, bins that end inside
    a wave's 16 (F = 17, 65, 129) and whole waves without a bin (F = 65: three of four).  (The
    register-fed passes these were compared with in round 5 serve the frame counts off the tile
    grid, test_fast_gauss_mnmf_general_shapes_against_oracle.)"""
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    X = np.stack([nmf_mixture(500 + b, M, F, T) for b in range(B)])
    rng = np.random.default_rng(3)
    kw = dict(basis=rng.random((B, N, F, K)), activation=rng.random((B, N, K, T)),
              spatial=rng.random((B, F, N, M)))
    if B == 1:
        X, kw = X[0], {k: v[0] for k, v in kw.items()}
    m1 = FastGaussMNMF(n_basis=K, n_sources=N)
    m1(X, n_iter=3, **{k: v.copy() for k, v in kw.items()})
    assert m1._handover is not None
    for a in _fastmnmf_states(m1):
        assert np.isfinite(a).all()
    b0 = 0 if B == 1 else B - 1
    Xo = X if B == 1 else X[b0]
    ref = FastGaussMNMFOracle(n_basis=K, n_sources=N)
    Yr = ref.run(Xo, n_iter=3, **{k: (v if B == 1 else v[b0]).copy() for k, v in kw.items()})
    loss = np.asarray(m1.loss)
    np.testing.assert_allclose(loss if B == 1 else loss[:, b0], ref.loss, rtol=LOSS_RTOL)
    q = np.asarray(m1.diagonalizer)
    assert rel_err(q if B == 1 else q[b0], ref.diagonalizer) < TOL
    y = np.asarray(m1.output)
    assert rel_err(y if B == 1 else y[b0], Yr) < 1e-7


def test_aux_iva_ip1_resident_loss_loop_equals_reference_loop():
    """AuxLaplaceIVA-IP1, record_loss=True without callbacks: the loss of every state comes from the
    frame powers the next iteration forms anyway and the list from one download; with a callback the
    reference's loop runs (compute_loss() per iteration).  Same list, same filters."""
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, N, F, T = 3, 4, 33, 70
    X = np.stack([nmf_mixture(800 + b, N, F, T) for b in range(B)])
    for algo in ("IP", "IP1"):
        for initial_call in (True, False):
            m1 = AuxLaplaceIVA(spatial_algorithm=algo)
            Y1 = m1(X, n_iter=5, initial_call=initial_call)
            m2 = AuxLaplaceIVA(spatial_algorithm=algo, callbacks=lambda method: None)
            Y2 = m2(X, n_iter=5, initial_call=initial_call)
            assert len(m1.loss) == len(m2.loss) == (6 if initial_call else 5)
            # (the two loops launch the same state kernels: frame powers are summed without atomics)
            np.testing.assert_allclose(np.asarray(m1.loss), np.asarray(m2.loss), rtol=1e-12)
            assert np.array_equal(Y1, Y2)
            assert np.array_equal(m1.demix_filter, m2.demix_filter)


@pytest.mark.parametrize("N,T", [(2, 40), (4, 300), (8, 70)])
def test_aux_iva_iss_tracked_logdet_equals_rebuilt_filters(N, T):
    """ISS keeps no filters; compute_loss() needs sum_i log|det W_i|.  The fused sweep kernel tracks
    it (each sweep multiplies det W_i by d_in^(-1/2)); without the tracker W is rebuilt from Y X^H
    for every recorded loss.  Same loss lists, also from a non-identity initial filter."""
    from ssspy_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, F = 2, 33
    X = np.stack([nmf_mixture(850 + b, N, F, T) for b in range(B)])
    rng = np.random.default_rng(N)
    W0 = np.eye(N) + 0.2 * (rng.standard_normal((B, F, N, N)) + 1j * rng.standard_normal((B, F, N, N)))
    for cls in (AuxLaplaceIVA, AuxGaussIVA):
        class Untracked(cls):
            def _tracked_logdet(self):
                return None

        for kw in ({}, {"demix_filter": W0}):
            m1 = cls(spatial_algorithm="ISS")
            m1(X, n_iter=4, **{k: v.copy() for k, v in kw.items()})
            assert m1._tracked_logdet() is None  # retired by the scale restoration at the end
            m2 = Untracked(spatial_algorithm="ISS")
            m2(X, n_iter=4, **{k: v.copy() for k, v in kw.items()})
            np.testing.assert_allclose(np.asarray(m1.loss), np.asarray(m2.loss), rtol=1e-10)


@pytest.mark.parametrize("N,norm", [(2, True), (4, True), (4, False), (8, True), (3, "projection_back")])
def test_ilrma_iss_tracked_logdet_equals_rebuilt_filters(N, norm):
    """ILRMA on the ISS state: the sweeps and the power normalisation move the tracked
    sum_i log|det W_i| along; without it (and with the projection-back normalisation, which retires
    it) compute_loss() rebuilds W from Y X^H.  Same loss lists."""
    from ssspy_amd.bss.ilrma import GaussILRMA, TILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, F, T, K = 2, 33, 70, 4
    X = np.stack([nmf_mixture(870 + b, N, F, T) for b in range(B)])
    rng = np.random.default_rng(N)
    init = dict(basis=rng.random((B, N, F, K)), activation=rng.random((B, N, K, T)))
    for cls, extra in ((GaussILRMA, {}), (TILRMA, {"dof": 5.0})):
        class Untracked(cls):
            def _tracked_logdet(self):
                return None

        kw = dict(n_basis=K, spatial_algorithm="ISS", normalization=norm, **extra)
        m1 = cls(**kw)
        m1(X, n_iter=4, **{k: v.copy() for k, v in init.items()})
        m2 = Untracked(**kw)
        m2(X, n_iter=4, **{k: v.copy() for k, v in init.items()})
        np.testing.assert_allclose(np.asarray(m1.loss), np.asarray(m2.loss), rtol=1e-10)
        assert rel_err(m1.output, m2.output) < 1e-12


def test_stream_handle_follows_the_current_stream():
    """The raw-handle fast path must name the stream torch would launch on, also inside a
    `torch.cuda.stream(...)` context (kernels enqueue on it; a wrong handle would race)."""
    import torch
    from ssspy_amd import _device as dv

    assert dv.stream_handle() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert dv.stream_handle() == side.cuda_stream
        assert dv.stream_handle() == torch.cuda.current_stream().cuda_stream
    assert dv.stream_handle() == torch.cuda.current_stream().cuda_stream


def test_iss_logdet_tracker_only_with_record_loss():
    """The tracked sweep kernel is a separate, slightly slower instantiation (the N = 8 slab kernel
    sits at the register cap): it must run only when the loss is recorded."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(5, 3, 17, 40)
    for record_loss in (False, True):
        for m in (AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=record_loss),
                  GaussILRMA(n_basis=2, spatial_algorithm="ISS", record_loss=record_loss)):
            m._bind_input(X)
            if isinstance(m, AuxLaplaceIVA):
                from ssspy_amd.bss.iva import _device_contrast
                m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
                m._reset()
            else:
                m._reset(flooring_fn=m.flooring_fn)
            m.update_once()
            assert (m._tracked_logdet() is not None) == record_loss


def test_fast_gauss_mnmf_resident_loss_loop_equals_reference_loop():
    """record_loss=True without callbacks keeps the loss terms in HBM until the end of __call__;
    with a callback the reference's loop (compute_loss() and a download per iteration) runs.  Same
    list, same state; initial_call=False drops the first entry in both."""
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    B, M, F, T, K = 2, 4, 40, 64, 5
    X = np.stack([nmf_mixture(300 + b, M, F, T) for b in range(B)])
    kw = dict(basis=np.random.default_rng(1).random((B, M, F, K)),
              activation=np.random.default_rng(2).random((B, M, K, T)),
              spatial=np.random.default_rng(4).random((B, F, M, M)))
    for initial_call in (True, False):
        seen = []
        m1 = FastGaussMNMF(n_basis=K)
        Y1 = m1(X, n_iter=4, initial_call=initial_call, **{k: v.copy() for k, v in kw.items()})
        m2 = FastGaussMNMF(n_basis=K, callbacks=lambda method: seen.append(len(method.loss)))
        Y2 = m2(X, n_iter=4, initial_call=initial_call, **{k: v.copy() for k, v in kw.items()})
        assert len(m1.loss) == len(m2.loss) == (5 if initial_call else 4)
        assert seen == list(range(1, len(m2.loss) + 1)) if initial_call else len(seen) == 4
        np.testing.assert_allclose(np.asarray(m1.loss), np.asarray(m2.loss), rtol=1e-11)
        assert rel_err(Y1, Y2) < 1e-11 and rel_err(m1.diagonalizer, m2.diagonalizer) < 1e-11


def test_fast_gauss_mnmf_handover_follows_state_changes(monkeypatch):
    """The hand-over is dropped whenever the diagonaliser moves outside the spatial pass: caller
    assignment, single steps out of order, IP2; a batch keeps one scale per (mixture, channel)."""
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    B, M, F, T, K = 3, 4, 40, 64, 6
    X = np.stack([nmf_mixture(90 + b, M, F, T) for b in range(B)])
    kw = dict(basis=np.random.default_rng(1).random((B, M, F, K)),
              activation=np.random.default_rng(2).random((B, M, K, T)),
              spatial=np.random.default_rng(4).random((B, F, M, M)))

    def scenario(algo):
        m = FastGaussMNMF(n_basis=K, diagonalizer_algorithm=algo)
        m(X, n_iter=2, **{k: v.copy() for k, v in kw.items()})
        m.diagonalizer = np.array(m.diagonalizer) * (1.0 + 0.25j)      # caller moves Q
        m.update_once()
        m.update_once()
        m.update_basis()
        m.update_diagonalizer()                                        # Q moves, no spatial pass
        m.update_activation()
        m.update_spatial()
        m.normalize()
        m.update_basis()
        m.update_activation()
        m.spatial = np.array(m.spatial) * 1.5                          # D alone: hand-over stays
        m.update_once()
        return m, _fastmnmf_states(m)[:4]

    for algo in ("IP1", "IP2"):
        m1, with_handover = scenario(algo)
        assert m1._handover is not None
        with monkeypatch.context() as mp:
            mp.setitem(_routes.VALUES, "handover", False)
            m2, plain = scenario(algo)
            assert m2._handover is None
        for a, b in zip(with_handover, plain):
            # nine iterations of rounding-level differences (amplified by the pairwise eigenproblems)
            assert rel_err(a, b) < (1e-9 if algo == "IP1" else 1e-6)


def test_wiener_filter_floors_small_eigenvalues():
    """to_psd inside separate(): with a huge floor every eigenvalue is clamped, so R = eps I and the
    output is x-independent of the spatial model's conditioning: Y_n = R_n[ref,:] x / eps."""
    import functools
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.special.flooring import max_flooring
    from ssspy_amd.utils.dataset import iid_mixture

    M, F, T, K = 3, 9, 20, 2
    X = iid_mixture(3, M, F, T)
    kw = dict(basis=np.random.default_rng(1).random((M, F, K)),
              activation=np.random.default_rng(2).random((M, K, T)),
              spatial=np.random.default_rng(4).random((F, M, M)))
    ref = FastGaussMNMFOracle(n_basis=K, flooring=("max", 1e3))
    Yr = ref.run(X, n_iter=0, **{k: v.copy() for k, v in kw.items()})
    m = FastGaussMNMF(n_basis=K, flooring_fn=functools.partial(max_flooring, eps=1e3))
    Y = m(X, n_iter=0, **kw)
    assert rel_err(Y, Yr) < 1e-9


# ------------------------------------------------------------------------------- linalg / special
def test_inv2_against_golden_and_identity():
    from ssspy_amd.linalg import inv2

    g = load_golden("operators")
    assert rel_err(inv2(g["inv2_in"]), g["inv2_out"]) < 1e-13
    X = np.random.default_rng(0).standard_normal((5, 3, 2, 2))
    out = inv2(X)
    assert out.dtype == np.float64
    assert np.allclose(X @ out, np.eye(2), atol=1e-10)


@pytest.mark.parametrize("N", [2, 3, 4, 8, 9, 13, 16])
def test_solve_matches_numpy(N):
    from ssspy_amd.linalg import solve

    rng = np.random.default_rng(N)
    A = rng.standard_normal((7, 5, N, N)) + 1j * rng.standard_normal((7, 5, N, N))
    b = rng.standard_normal((7, 5, N)) + 1j * rng.standard_normal((7, 5, N))
    Bm = rng.standard_normal((7, 5, N, 3)) + 1j * rng.standard_normal((7, 5, N, 3))
    assert rel_err(solve(A, b), np.linalg.solve(A, b[..., None])[..., 0]) < 1e-11
    assert rel_err(solve(A, Bm), np.linalg.solve(A, Bm)) < 1e-11
    with pytest.raises(np.linalg.LinAlgError):
        solve(np.zeros((2, N, N), dtype=complex), np.ones((2, N), dtype=complex))


@pytest.mark.parametrize("M", [2, 3, 4, 6, 7, 8, 9, 12, 16])
def test_eigh_properties(M):
    """The reference's own property checks (tests/package/linalg/test_eigh.py): A z = lamb z,
    ascending eigenvalues; plus agreement with LAPACK eigenvalues and unitarity."""
    from ssspy_amd.linalg import eigh

    rng = np.random.default_rng(10 + M)
    A = rng.standard_normal((4, 9, M, M)) + 1j * rng.standard_normal((4, 9, M, M))
    A = A @ A.swapaxes(-2, -1).conj() - 0.3 * np.eye(M)
    lamb, V = eigh(A)
    assert np.all(np.diff(lamb, axis=-1) >= 0)
    assert rel_err(lamb, np.linalg.eigvalsh(A)) < 1e-12
    assert rel_err(A @ V, V * lamb[..., None, :]) < 1e-12
    assert rel_err(V.swapaxes(-2, -1).conj() @ V, np.broadcast_to(np.eye(M), V.shape)) < 1e-12


@pytest.mark.parametrize("type", [1, 2, 3])
def test_eigh2_generalised(type):
    from ssspy_amd.linalg import eigh2

    g = load_golden("operators")
    A, B = g["eigh2_A"], g["eigh2_B"]
    lamb, z = eigh2(A, B, type=type)
    zk = z  # columns are eigenvectors
    if type == 1:
        assert rel_err(A @ zk, (B @ zk) * lamb[..., None, :]) < 1e-11
        assert rel_err(lamb, g["eigh2_lamb"]) < 1e-12
        # the reference's eigenvectors WITH their phases: its eigh2 takes them from np.linalg.eigh on
        # the 2 x 2 matrix C (ssspy/linalg/eigh.py:198), and csrc/eigh2.hpp restates what LAPACK does
        # there (round 6; before: equal up to a per-column phase)
        assert rel_err(zk, g["eigh2_z"]) < 1e-10
    elif type == 2:
        assert rel_err(A @ B @ zk, zk * lamb[..., None, :]) < 1e-11
    else:
        assert rel_err(B @ A @ zk, zk * lamb[..., None, :]) < 1e-11
    assert np.all(lamb[..., 0] <= lamb[..., 1])
    # every type, and the plain 2 x 2 problem, against NumPy's route (Cholesky + np.linalg.eigh) with
    # the phases: real and complex, an off-diagonal entry that is real, imaginary, zero
    rng = np.random.default_rng(type)
    n = 200
    X = rng.standard_normal((n, 2, 6)) + 1j * rng.standard_normal((n, 2, 6))
    Y = rng.standard_normal((n, 2, 6)) + 1j * rng.standard_normal((n, 2, 6))
    A2, B2 = X @ X.swapaxes(-2, -1).conj(), Y @ Y.swapaxes(-2, -1).conj()
    A2[:40] = A2[:40].real                    # real symmetric problems
    B2[:20] = B2[:20].real
    A2[40:50, 0, 1] = A2[40:50, 0, 1].imag * 1j   # purely imaginary coupling
    A2[40:50, 1, 0] = A2[40:50, 0, 1].conj()
    A2[50:55, 0, 1] = A2[50:55, 1, 0] = 0.0   # already diagonal
    L = np.linalg.cholesky(B2)
    LH = L.swapaxes(-2, -1).conj()
    Li = np.linalg.inv(L)
    C = Li @ A2 @ Li.swapaxes(-2, -1).conj() if type == 1 else LH @ A2 @ L
    lam_r, y = np.linalg.eigh(C)
    z_r = L @ y if type == 3 else np.linalg.inv(LH) @ y
    lam2, z2 = eigh2(A2, B2, type=type)
    assert rel_err(lam2, lam_r) < 1e-11
    assert rel_err(z2, z_r) < 1e-9
    lam_p, V_p = eigh2(A2)
    lam_n, V_n = np.linalg.eigh(A2)
    assert rel_err(lam_p, lam_n) < 1e-12 and rel_err(V_p, V_n) < 1e-10


@pytest.mark.parametrize("N", [2, 3, 4, 8])
def test_to_psd_against_golden(N):
    from ssspy_amd.special import to_psd

    g = load_golden("operators")
    out = to_psd(g["psd_n{}_in".format(N)])
    assert rel_err(out, g["psd_n{}_out".format(N)]) < 1e-11
    assert np.all(np.linalg.eigvalsh(out) > 0)


# ------------------------------------------------------------------------------- pairwise operators
@pytest.mark.parametrize("N", [2, 3, 4, 8])
def test_pairwise_operators_against_oracle(N):
    from oracle import spatial as sp
    from ssspy_amd.bss._update_spatial_model import update_by_ip2, update_by_iss2
    from ssspy_amd.utils.select_pair import combination_pair_selector

    g = load_golden("operators")
    W, U = g["ip1_n{}_W".format(N)], g["ip1_n{}_U".format(N)]
    out = update_by_ip2(W.copy(), U)
    ref = sp.update_by_ip2(W, U)
    assert pair_err(out, ref, "demix_filter", N) < 1e-9
    out = update_by_ip2(W.copy(), U, pair_selector=combination_pair_selector)
    ref = sp.update_by_ip2(W, U, pairs=list(combination_pair_selector(N)))
    assert pair_err(out, ref, "demix_filter", N) < 1e-9
    Y, varphi = g["iss1_n{}_Y".format(N)], g["iss1_n{}_varphi".format(N)]
    out = update_by_iss2(Y, varphi)
    ref = sp.update_by_iss2(Y, varphi)
    assert pair_err(out, ref, "output", N) < 1e-9
    # negative indices wrap, as in the reference
    out = update_by_iss2(Y, varphi[:, :1, :], pair_selector=lambda n: [(-1, 0)])
    ref = sp.update_by_iss2(Y, varphi[:, :1, :], pairs=[(N - 1, 0)])
    assert pair_err(out, ref, "output", N) < 1e-9


MNMF_IP2_CASES = ["fmnmf_ip2_m2", "fmnmf_ip2_m3", "fmnmf_ip2_m4", "fmnmf_ip2_m3_n2",
                  "fmnmf_ip2_m4_comb", "fmnmf_ip2_m5"]


@pytest.mark.parametrize("case", MNMF_IP2_CASES)
def test_fast_gauss_mnmf_ip2_against_golden(case):
    """diagonalizer_algorithm="IP2" against fixtures generated by the reference
    (ssspy/bss/mnmf.py:1516-1633; sequential and combination pair selectors, n_sources < n_channels,
    5 channels).  The rows of Q inherit the arbitrary phase of the 2x2 generalised eigenvectors,
    so Q is compared up to one phase per (bin, row) and |Q x| directly; D, T, V, the loss list and
    the Wiener-filter output do not depend on that phase."""
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.select_pair import combination_pair_selector

    g = load_golden(case)
    extra = {}
    if str(g["meta_pairs"]) == "combination":
        extra["pair_selector"] = combination_pair_selector
    snap = Snap(["diagonalizer", "spatial", "basis", "activation"])
    m = FastGaussMNMF(n_basis=int(g["meta_n_basis"]), n_sources=int(g["meta_n_sources"]),
                      diagonalizer_algorithm="IP2", flooring_fn=_flooring_fn(g), callbacks=snap,
                      normalization=_option(g["meta_normalization"]), **extra)
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]), basis=g["basis0"], activation=g["activation0"],
          spatial=g["spatial0"].copy())
    checked = 0
    Xt = g["X"].transpose(1, 0, 2)
    for key, value in snap.store.items():
        assert key in g, key
        if key.endswith("_diagonalizer"):
            assert rel_err_up_to_phase(value, g[key], "demix_filter") < TOL, key
            assert rel_err(np.abs(value @ Xt), np.abs(g[key] @ Xt)) < TOL, key  # |Q x|
            # the rows WITH the phases the reference's np.linalg.eigh leaves (round 6, eigh2.hpp),
            # from 3 channels on (2 channels: the pair is visited twice, see pair_err)
            assert pair_err(value, g[key], "demix_filter", g["X"].shape[0]) < 1e-6, key
        else:
            assert rel_err(value, g[key]) < TOL, key
        checked += 1
    assert checked >= 8
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    assert rel_err(m.spatial, g["final_spatial"]) < TOL
    assert rel_err(m.basis, g["final_basis"]) < TOL
    assert rel_err(m.activation, g["final_activation"]) < TOL
    assert rel_err_up_to_phase(m.diagonalizer, g["final_diagonalizer"], "demix_filter") < TOL
    assert pair_err(m.diagonalizer, g["final_diagonalizer"], "demix_filter", g["X"].shape[0]) < 1e-6
    assert rel_err(Y, g["final_output"]) < 1e-7  # Wiener filter: eigh + solve, cond(R)-amplified


@pytest.mark.parametrize("N", [2, 3, 4, 5, 8])
def test_pairwise_operators_against_golden(N):
    """update_by_ip2 / update_by_iss2 against the reference's own outputs
    (ssspy/bss/_update_spatial_model.py:81-143, 197-314): default pairs, every combination, explicit
    lists with negative / descending indices (:241-251), broadcast weights, add-flooring,
    overwrite semantics."""
    import functools

    from ssspy_amd.bss._update_spatial_model import update_by_ip2, update_by_iss2
    from ssspy_amd.special.flooring import add_flooring
    from ssspy_amd.utils.select_pair import combination_pair_selector

    g = load_golden("pairwise_operators")
    p = lambda s: g["n{}_".format(N) + s]  # noqa: E731
    W, U, Y, varphi = p("W"), p("U"), p("Y"), p("varphi")
    add = functools.partial(add_flooring, eps=1e-3)
    tol = 1e-9
    Wc = W.copy()
    out = update_by_ip2(Wc, U)
    assert out is Wc  # overwrite=True aliases, as in the reference
    assert pair_err(out, p("ip2_out"), "demix_filter", N) < tol
    Wc = W.copy()
    out = update_by_ip2(Wc, U, overwrite=False)
    assert out is not Wc and np.array_equal(Wc, W)
    assert pair_err(out, p("ip2_out_copy"), "demix_filter", N) < tol
    out = update_by_ip2(W.copy(), U, pair_selector=combination_pair_selector)
    assert pair_err(out, p("ip2_out_comb"), "demix_filter", N) < tol
    out = update_by_ip2(W.copy(), U, flooring_fn=add)
    assert pair_err(out, p("ip2_out_add"), "demix_filter", N) < tol
    pairs = [tuple(int(v) for v in pr) for pr in p("ip2_pairs")]
    out = update_by_ip2(W.copy(), U, pair_selector=lambda n: pairs)
    # (this list visits one pair twice in a row: the second visit's phase is rounding noise)
    assert rel_err_up_to_phase(out, p("ip2_out_pairs"), "demix_filter") < tol
    Yc = Y.copy()
    out = update_by_iss2(Yc, varphi)
    assert pair_err(out, p("iss2_out"), "output", N) < tol
    out = update_by_iss2(Y.copy(), varphi, pair_selector=combination_pair_selector)
    assert pair_err(out, p("iss2_out_comb"), "output", N) < tol
    out = update_by_iss2(Y.copy(), varphi[:, :1, :], flooring_fn=add)
    assert pair_err(out, p("iss2_out_bcast_add"), "output", N) < tol
    pairs = [tuple(int(v) for v in pr) for pr in p("iss2_pairs")]
    out = update_by_iss2(Y.copy(), varphi, pair_selector=lambda n: pairs)
    assert pair_err(out, p("iss2_out_pairs"), "output", N) < tol


# ------------------------------------------------------------------------------- full BASELINE sizes
def test_aux_iva_iss_config3_full_size_properties():
    """BASELINE.json configs[2] (AuxLaplaceIVA-ISS, N=8, F=2049, T=1024): size-independent
    properties -- the auxiliary-function loss never increases, the fused ISS kernel's state stays a
    linear transform of the input (Y_i = W_i X_i with the least-squares W_i reproduces Y), and
    projection back makes the reference channel reconstruct exactly (sum_n y_n = x_ref)."""
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T = 8, 2049, 1024
    X = nmf_mixture(3000, N, F, T)
    m = AuxLaplaceIVA(spatial_algorithm="ISS")
    Y = m(X, n_iter=12)
    loss = np.array(m.loss)
    assert np.all(np.diff(loss) <= 1e-9 * np.abs(loss[:-1]))
    assert m.demix_filter is None and Y.shape == X.shape
    assert rel_err(Y.sum(axis=0), X[0]) < 1e-9
    # linearity per bin: least-squares filter from (Y, X) reproduces Y on a sample of bins
    for i in (0, 1024, 2048):
        Xi, Yi = X[:, i, :], Y[:, i, :]
        Wi = Yi @ Xi.conj().T @ np.linalg.inv(Xi @ Xi.conj().T)
        assert rel_err(Wi @ Xi, Yi) < 1e-9


def test_fast_gauss_mnmf_config4_full_size_properties():
    """BASELINE.json configs[3] (FastGaussMNMF, N=M=4, F=1025, T=512, K=8): the loss is non-increasing
    over the iterations (MM + IP guarantee), the power normalisation holds
    (mean_{i,j} |q_m^H x|^2 = 1), the Wiener filter outputs sum to the reference channel, and the
    first iterations match the oracle."""
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    M, F, T, K = 4, 1025, 512, 8
    X = nmf_mixture(4000, M, F, T)
    kw = dict(basis=np.random.default_rng(1).random((M, F, K)),
              activation=np.random.default_rng(2).random((M, K, T)),
              spatial=np.random.default_rng(4).random((F, M, M)))
    m = FastGaussMNMF(n_basis=K)
    Y = m(X, n_iter=15, **kw)
    loss = np.array(m.loss)
    assert np.all(np.diff(loss) <= 1e-9 * np.abs(loss[:-1]))
    QX = m.diagonalizer @ X.transpose(1, 0, 2)
    np.testing.assert_allclose(np.mean(np.abs(QX) ** 2, axis=(0, 2)), 1.0, rtol=1e-10)
    assert rel_err(Y.sum(axis=0), X[0]) < 1e-8  # sum_n W_n = R^-1 sum_n R_n = I on the reference row
    ref = FastGaussMNMFOracle(n_basis=K, record_loss=True)
    ref.reset(X, **{k: v.copy() for k, v in kw.items()})
    ref_loss = [ref.compute_loss()]
    ref.update_once()
    ref_loss.append(ref.compute_loss())
    np.testing.assert_allclose(loss[:2], ref_loss, rtol=LOSS_RTOL)


# ------------------------------------------------------------------------------- large-batch code paths
@pytest.mark.parametrize("T", [32, 33])
def test_large_batch_paths_against_oracle(T):
    """With >= 512 workgroups per launch the kernels switch to their large-batch form (no frame
    chunking in the ILRMA fast path, bin-split FastMNMF kernels, AuxIVA's tuned covariance walk) -- the one bench.py times.  600 tiny
    mixtures; first, middle and last are checked against the oracle.  T = 33: rows of an odd length
    (activation rows that start 8 bytes off a 16-byte boundary, last row ending the buffer)."""
    from oracle.ilrma import GaussILRMAOracle
    from oracle.iva import AuxIVAOracle
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    B, N, F, K = 600, 3, 18, 4
    rng = np.random.default_rng(5)
    X = np.stack([nmf_mixture(7000 + b, N, F, T) for b in range(B)])
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    spatial = rng.random((B, F, N, N))
    m = GaussILRMA(n_basis=K)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    mm = FastGaussMNMF(n_basis=K)
    Ym = mm(X, n_iter=3, basis=basis, activation=act, spatial=spatial)
    mi = AuxLaplaceIVA(spatial_algorithm="IP")  # frame-weighted covariance, tuned tile walk
    Yi = mi(X, n_iter=3)
    for b in (0, 301, B - 1):
        refi = AuxIVAOracle(spatial_algorithm="IP", contrast="laplace")
        Yri = refi.run(X[b], n_iter=3)
        assert rel_err(Yi[b], Yri) < TOL
        np.testing.assert_allclose(np.asarray(mi.loss)[:, b], refi.loss, rtol=LOSS_RTOL)
        ref = GaussILRMAOracle(n_basis=K)
        Yr = ref.run(X[b], n_iter=3, basis=basis[b], activation=act[b])
        assert rel_err(Y[b], Yr) < TOL
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref.loss, rtol=LOSS_RTOL)
        refm = FastGaussMNMFOracle(n_basis=K)
        Yrm = refm.run(X[b], n_iter=3, basis=basis[b], activation=act[b], spatial=spatial[b].copy())
        np.testing.assert_allclose(np.asarray(mm.loss)[:, b], refm.loss, rtol=LOSS_RTOL)
        assert rel_err(mm.diagonalizer[b], refm.diagonalizer) < TOL
        assert rel_err(Ym[b], Yrm) < 1e-7


# ------------------------------------------------------------------------------- STFT / ISTFT
@pytest.mark.parametrize("n_fft,hop,L", [(64, 16, 1000), (256, 128, 4097), (1024, 256, 9000),
                                         (4096, 1024, 20000), (128, 128, 777), (8192, 2048, 40000)])
def test_stft_istft_against_scipy(n_fft, hop, L):
    """The transforms the reference's workflow takes from scipy.signal, with SciPy's defaults."""
    import scipy.signal as ss

    from ssspy_amd.transform import istft, stft

    rng = np.random.default_rng(n_fft + hop)
    x = rng.standard_normal((3, L))
    _, _, Zr = ss.stft(x, window="hann", nperseg=n_fft, noverlap=n_fft - hop)
    Z = stft(x, n_fft=n_fft, hop_length=hop)
    assert Z.shape == Zr.shape and Z.dtype == np.complex128
    assert rel_err(Z, Zr) < 1e-12
    _, yr = ss.istft(Zr, window="hann", nperseg=n_fft, noverlap=n_fft - hop)
    y = istft(Zr, n_fft=n_fft, hop_length=hop)
    assert y.shape == yr.shape
    assert rel_err(y, yr) < 1e-12
    if hop <= n_fft // 2:  # NOLA holds: perfect reconstruction
        assert rel_err(y[:, :L], x) < 1e-12


@pytest.mark.parametrize("n_fft,hop,window", [
    (1000, 250, "hann"), (1536, 384, "hamming"), (300, 75, ("kaiser", 8.6)), (4096 - 2, 1000, "blackman"),
    (441, 147, "hann"), (25, 5, ("tukey", 0.5)), (2, 1, "boxcar"), (3000, 1000, "bartlett")])
def test_stft_any_length_and_named_windows_against_scipy(n_fft, hop, window):
    """SciPy takes any nperseg (even or odd) and any of its named windows; here lengths that are not a
    power of two go through Bluestein's chirp-z on two radix-2 transforms in LDS."""
    import scipy.signal as ss

    from ssspy_amd.transform import istft, stft

    rng = np.random.default_rng(n_fft)
    L = 7 * n_fft + 13
    x = rng.standard_normal((2, L))
    _, _, Zr = ss.stft(x, window=window, nperseg=n_fft, noverlap=n_fft - hop)
    Z = stft(x, n_fft=n_fft, hop_length=hop, window=window)
    assert Z.shape == Zr.shape
    assert rel_err(Z, Zr) < 1e-11
    _, yr = ss.istft(Zr, window=window, nperseg=n_fft, noverlap=n_fft - hop)
    y = istft(Zr, n_fft=n_fft, hop_length=hop, window=window)
    assert y.shape == yr.shape
    assert rel_err(y, yr) < 1e-11


def test_stft_of_a_signal_shorter_than_the_window_follows_scipy():
    """scipy.signal.stft shrinks nperseg to the signal's length (with a warning) and keeps noverlap,
    raising when the overlap no longer fits (scipy/signal/_spectral_py.py, _triage_segments); found
    by benchmarks/fuzz_transform.py."""
    import scipy.signal as ss

    from ssspy_amd.transform import stft

    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 193))
    with pytest.warns(UserWarning, match="nperseg = 256 is greater than input length"):
        _, _, Zr = ss.stft(x, window="hann", nperseg=256, noverlap=128)
    with pytest.warns(UserWarning, match="nperseg = 256 is greater than input length"):
        Z = stft(x, n_fft=256, hop_length=128)
    assert Z.shape == Zr.shape == (2, 97, 4)
    assert rel_err(Z, Zr) < 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ValueError, match="noverlap must be less than nperseg"):
            ss.stft(x[:, :100], window="hann", nperseg=256, noverlap=128)
        with pytest.raises(ValueError, match="noverlap must be less than nperseg"):
            stft(x[:, :100], n_fft=256, hop_length=128)


@pytest.mark.parametrize("n_fft,hop,window", [(5000, 1250, "hann"), (6001, 2000, "hamming"),
                                             (16384, 4096, "hann"), (8191, 2048, "blackman")])
def test_stft_beyond_the_lds_against_scipy(n_fft, hop, window):
    """Round 6: transforms of more than 8192 points (a power of two above 8192, any other length above
    4096 through Bluestein) run on workgroup-private slices of the workspace in HBM -- SciPy takes
    any nperseg (round-5 verdict, missing item 4).  More segments than resident workgroups too."""
    import scipy.signal as ss

    from ssspy_amd.transform import istft, stft

    rng = np.random.default_rng(n_fft)
    L = (140 if n_fft == 5000 else 9) * hop + 13  # (5000: 2 x ~140 segments > 256 workgroups)
    x = rng.standard_normal((2, L))
    _, _, Zr = ss.stft(x, window=window, nperseg=n_fft, noverlap=n_fft - hop)
    Z = stft(x, n_fft=n_fft, hop_length=hop, window=window)
    assert Z.shape == Zr.shape
    assert rel_err(Z, Zr) < 1e-11
    _, yr = ss.istft(Zr, window=window, nperseg=n_fft, noverlap=n_fft - hop)
    y = istft(Zr, n_fft=n_fft, hop_length=hop, window=window)
    assert y.shape == yr.shape
    assert rel_err(y, yr) < 1e-11


def test_stft_rejects_what_it_cannot_hold():
    from ssspy_amd.transform import stft

    with pytest.raises(NotImplementedError):
        stft(np.zeros((1, 200000)), n_fft=40000)  # Bluestein needs 131072 points: above the plan's 65536
    with pytest.raises(ValueError, match="Unknown window"):
        stft(np.zeros((1, 2000)), n_fft=64, window="no_such_window")


def test_waveform_to_waveform_stays_on_device():
    """stft -> separator -> istft with device tensors between the stages == the SciPy / NumPy path."""
    import scipy.signal as ss
    import torch

    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.transform import istft, stft

    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6000))
    Zd = stft(x, n_fft=256, hop_length=64, device_output=True)
    assert isinstance(Zd, torch.Tensor) and Zd.is_cuda
    basis, act = rng.random((2, 129, 3)), rng.random((2, 3, Zd.shape[-1]))
    m = GaussILRMA(n_basis=3)
    m(Zd, n_iter=3, basis=basis, activation=act)
    yd = istft(m._state_dev("output")[0], n_fft=256, hop_length=64, device_output=True)
    assert yd.is_cuda
    # the same through SciPy and host arrays
    _, _, Zr = ss.stft(x, window="hann", nperseg=256, noverlap=192)
    Y2 = GaussILRMA(n_basis=3)(Zr, n_iter=3, basis=basis, activation=act)
    _, yr = ss.istft(Y2, window="hann", nperseg=256, noverlap=192)
    assert rel_err(yd.cpu().numpy(), yr) < 1e-9


# ------------------------------------------------------------ Hermitian matrix functions (linalg)
def _psd(rng, lead, M, T=32, complex_=True):
    x = rng.standard_normal(lead + (M, T))
    if complex_:
        x = x + 1j * rng.standard_normal(lead + (M, T))
    return np.mean(x[..., :, None, :] * x[..., None, :, :].conj(), axis=-1)


@pytest.mark.parametrize("M", [3, 4, 6, 7, 8, 10, 16])
@pytest.mark.parametrize("type", [1, 2, 3])
def test_generalized_eigh(M, type):
    """The reference's property checks for the generalised problem (tests/package/linalg/test_eigh.py),
    beyond 2 x 2; eigenvalues also against the Cholesky route in NumPy."""
    from ssspy_amd.linalg import eigh

    rng = np.random.default_rng(111 + M)
    A, B = _psd(rng, (5,), M), _psd(rng, (5,), M)
    lamb, z = eigh(A, B, type=type)
    assert lamb.shape == (5, M) and z.shape == (5, M, M)
    assert np.all(np.diff(lamb, axis=-1) >= 0)
    if type == 1:
        lhs, rhs = A @ z, lamb[:, None, :] * (B @ z)
    elif type == 2:
        lhs, rhs = A @ B @ z, lamb[:, None, :] * z
    else:
        lhs, rhs = B @ A @ z, lamb[:, None, :] * z
    assert rel_err(lhs, rhs) < 1e-11
    L = np.linalg.cholesky(B)
    Li = np.linalg.inv(L)
    C = Li @ A @ Li.swapaxes(-2, -1).conj() if type == 1 else L.swapaxes(-2, -1).conj() @ A @ L
    np.testing.assert_allclose(lamb, np.linalg.eigvalsh(C), rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("M", [3, 4, 6, 7, 8, 10, 16])
@pytest.mark.parametrize("is_complex", [True, False])
def test_sqrtmh_invsqrtmh(M, is_complex):
    """tests/package/linalg/test_sqrtm.py on the device."""
    from ssspy_amd.linalg import invsqrtmh, sqrtmh
    from ssspy_amd.special.flooring import max_flooring

    rng = np.random.default_rng(0)
    X = _psd(rng, (2,), M, complex_=is_complex)
    S = sqrtmh(X)
    assert S.dtype == X.dtype and rel_err(S @ S, X) < 1e-12
    for flooring_fn in (None, functools.partial(max_flooring, eps=1e-10)):
        Xi = invsqrtmh(X, flooring_fn=flooring_fn)
        assert rel_err(np.linalg.inv(Xi) @ np.linalg.inv(Xi), X) < 1e-10


@pytest.mark.parametrize("type", [1, 2, 3])
def test_gmeanmh(type):
    """tests/package/linalg/test_gmean.py: the Riccati property of each type, same data."""
    from ssspy_amd.linalg import gmeanmh

    rng = np.random.default_rng(0)
    size = (16, 32, 4, 1)

    def create_psd():
        x = rng.random(size) + 1j * rng.random(size)
        return np.mean(x * x.transpose(0, 1, 3, 2).conj(), axis=0)

    A, B = create_psd(), create_psd()
    G = gmeanmh(A, B, type=type)
    if type == 1:
        assert np.allclose(G @ np.linalg.inv(A) @ G, B)
    elif type == 2:
        assert np.allclose(G @ A @ G, B)
    else:
        assert np.allclose(G @ np.linalg.inv(A) @ G, np.linalg.inv(B))
    assert rel_err(G, G.swapaxes(-2, -1).conj()) < 1e-13


@pytest.mark.parametrize("M", [6, 7, 8, 9, 11, 16])
def test_hermitian_operators_at_6_to_16_channels_against_lapack(M):
    """eigh, to_psd, the generalised eigenproblem, sqrtmh / invsqrtmh and gmeanmh at 6 x 6 (a lane
    per matrix), 7 x 7 / 8 x 8 (a matrix on 8 lanes, hermitian_rows.hip) and 9 x 9 .. 16 x 16 (the
    size at run time, hermitian_rt.hip) against LAPACK and the defining identities; eigenvectors
    through the projectors z z^H (free of the phase each decomposition leaves)."""
    from ssspy_amd.linalg import eigh, gmeanmh, invsqrtmh, sqrtmh
    from ssspy_amd.special.flooring import max_flooring
    from ssspy_amd.special.psd import to_psd

    rng = np.random.default_rng(500 + M)
    lead = (70,)  # (more than two blocks of 32 matrices, a ragged last one)
    A, B = _psd(rng, lead, M), _psd(rng, lead, M)
    Hm = rng.standard_normal(lead + (M, M)) + 1j * rng.standard_normal(lead + (M, M))
    Hm = Hm + Hm.swapaxes(-2, -1).conj()  # indefinite
    floor = functools.partial(max_flooring, eps=0.5)
    rows = {"eigh": eigh(Hm), "psd": to_psd(Hm, flooring_fn=floor), "sqrt": sqrtmh(A),
            "invsqrt": invsqrtmh(A, flooring_fn=floor)}
    for t in (1, 2, 3):
        rows["gmean%d" % t] = gmeanmh(A, B, type=t)
        rows["geigh%d" % t] = eigh(A, B, type=t)

    def projectors(z):
        return z[..., :, None, :] * z[..., None, :, :].conj()  # [.., r, c, k] = z_rk conj(z_ck)

    lam0, V0 = np.linalg.eigh(Hm)
    assert rel_err(projectors(rows["eigh"][1]), projectors(V0)) < 1e-9
    import scipy.linalg

    for t in (1, 2, 3):  # generalised eigenvalues against scipy (LAPACK zhegv), one matrix pair each
        lam_t = rows["geigh%d" % t][0]
        for i in (0, 33, 69):
            np.testing.assert_allclose(lam_t[i], scipy.linalg.eigh(A[i], B[i], type=t, eigvals_only=True),
                                       rtol=1e-9)
    lamA, VA = np.linalg.eigh(A)  # (ssspy/linalg/sqrtm.py:54-64: the floor acts on sqrt(lambda))
    ref_invsqrt = (VA / np.maximum(np.sqrt(lamA), 0.5)[..., None, :]) @ VA.swapaxes(-2, -1).conj()
    assert rel_err(rows["invsqrt"], ref_invsqrt) < 1e-10
    assert rel_err(rows["gmean1"] @ np.linalg.inv(A) @ rows["gmean1"], B) < 1e-9
    assert rel_err(rows["gmean3"] @ np.linalg.inv(A) @ rows["gmean3"], np.linalg.inv(B)) < 1e-9
    # and against LAPACK where NumPy has the function
    np.testing.assert_allclose(rows["eigh"][0], np.linalg.eigvalsh(Hm), rtol=1e-11, atol=1e-12)
    lam, V = np.linalg.eigh(Hm)
    ref_psd = (V * np.maximum(lam, 0.5)[..., None, :]) @ V.swapaxes(-2, -1).conj()
    assert rel_err(rows["psd"], ref_psd) < 1e-11
    assert rel_err(rows["sqrt"] @ rows["sqrt"], A) < 1e-11
    assert rel_err(rows["gmean2"] @ A @ rows["gmean2"], B) < 1e-9


@pytest.mark.parametrize("L", [1, 2, 3, 5, 7, 8, 11, 15])
def test_lqpqm2_against_oracle(L):
    from oracle.ipa import lqpqm2 as oracle_lqpqm2
    from ssspy_amd.linalg import lqpqm2

    rng = np.random.default_rng(40 + L)
    n = 33
    H = _psd(rng, (n,), L, T=12)
    H = H / np.real(np.trace(H, axis1=-2, axis2=-1))[:, None, None]
    v = rng.standard_normal((n, L)) + 1j * rng.standard_normal((n, L))
    z = rng.random(n) * 2.0
    for max_iter in (1, 10):
        y = lqpqm2(H, v, z, max_iter=max_iter)
        yr = oracle_lqpqm2(H, v, z, ("max", 1e-10), max_iter)
        assert rel_err(y, yr) < 1e-10


@pytest.mark.parametrize("L", [6, 9, 15])
def test_lqpqm2_singular_fn_against_oracle(L):
    """singular_fn = default / None / a callable with problems whose v is exactly zero or small, up
    to the largest dimension (the run-time-L kernel from 8 on): the other problems as the oracle's
    (they alone decide the Newton step count), the singular ones in modulus."""
    from oracle.ipa import lqpqm2 as oracle_lqpqm2
    from test_oracle_golden import lqpqm_singular_check

    from ssspy_amd.linalg import lqpqm2

    rng = np.random.default_rng(140 + L)
    n = 70
    H = _psd(rng, (n,), L, T=3 * L)
    H = H / np.real(np.trace(H, axis1=-2, axis2=-1))[:, None, None]
    v = rng.standard_normal((n, L)) + 1j * rng.standard_normal((n, L))
    v[[0, 7, 64]] = 0.0
    v[[3, 65]] *= 1e-2
    z = rng.random(n) * 2.0
    norms = np.linalg.norm(v, axis=-1)
    for kw, singular in ((dict(), norms < 1e-10), (dict(singular_fn=None), norms == 0),
                         (dict(singular_fn=lambda x: x < 0.5), norms < 0.5)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = lqpqm2(H, v, z, **kw)
            ref = oracle_lqpqm2(H, v, z, ("max", 1e-10), 10, **kw)
        lqpqm_singular_check(y, np.asarray(ref), singular)


@pytest.mark.parametrize("L", [1, 2, 3, 5])
def test_lqpqm2_against_golden(L):
    """ssspy.linalg.lqpqm2 of the reference: the loop stops when every problem has converged (no
    warning at its default of ten steps) and warns when two steps are not enough."""
    import warnings

    from ssspy_amd.linalg import lqpqm2

    g = load_golden("ipa_operators")
    H, v, z = (g["lq{}_{}".format(L, k)] for k in "Hvz")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert rel_err(lqpqm2(H, v, z), g["lq{}_y".format(L)]) < 1e-10
    if L > 1:
        with pytest.warns(UserWarning, match="did not converge in 2 iterations"):
            y2 = lqpqm2(H, v, z, max_iter=2)
    else:
        y2 = lqpqm2(H, v, z, max_iter=2)
    assert rel_err(y2, g["lq{}_y_it2".format(L)]) < 1e-10


@pytest.mark.parametrize("L", [1, 2, 3, 5])
def test_lqpqm2_singular_fn_against_golden(L):
    """singular_fn = "flooring" (default), None and a callable, on problems whose v is exactly zero,
    below the floor, and small (ssspy/linalg/lqpqm.py:61-110).  Non-singular problems: equal to the
    reference.  Singular ones: the reference returns scale * (the last row of LAPACK's eigenvector
    matrix), every entry with the arbitrary phase of a different eigenvector -- the moduli are what
    is defined, and they are compared."""
    import warnings

    from test_oracle_golden import lqpqm_singular_check

    from ssspy_amd.linalg import lqpqm2

    g = load_golden("lqpqm_singular")
    H, v, z = (g["l{}_{}".format(L, k)] for k in ("H", "v", "z"))
    norms = np.linalg.norm(v, axis=-1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (convergence warnings are pinned by the test above)
        lqpqm_singular_check(lqpqm2(H, v, z), g["l{}_y_default".format(L)], norms < 1e-10)
        lqpqm_singular_check(lqpqm2(H, v, z, singular_fn=None), g["l{}_y_none".format(L)], norms == 0)
        lqpqm_singular_check(lqpqm2(H, v, z, singular_fn=lambda x: x < 0.5),
                             g["l{}_y_callable".format(L)], norms < 0.5)
        # no floor, three Newton steps.  The problems with ||v|| ~ 1e-13 that singular_fn=None sends
        # down the regular branch are left out: unfloored they divide quantities of order 1e-26 (the
        # reference itself returns NaN for two of them at L = 1), nothing there is pinned by anything.
        ref = g["l{}_y_none_nofloor_it3".format(L)]
        y = lqpqm2(H, v, z, flooring_fn=None, singular_fn=None, max_iter=3)
        ok = (norms == 0) | (norms > 1e-3)
        lqpqm_singular_check(y[ok], ref[ok], (norms == 0)[ok], tol=1e-9)


# ------------------------------------------------------------------------------- boundary (round 2)
@pytest.mark.parametrize("case", ["auxgeneric_ip1_n3", "auxgeneric_iss1_n2", "auxgeneric_ip2_n3"])
def test_generic_aux_iva_user_closures_against_golden(case):
    """The generic AuxIVA class with user closures (G_R(r) = r^1.5, not a built-in contrast):
    d_contrast_fn runs on the host on the (n_sources, n_frames) frame norms, contrast_fn on a host
    copy of the estimate when the loss is recorded; both passes over the spectrogram stay on the
    device.  ref: ssspy/bss/iva.py:1582-1635, :1785-1791, :2177-2192."""
    from ssspy_amd.bss.iva import AuxIVA

    g = load_golden(case)
    p = float(g["meta_power"])
    algo = str(g["meta_algo"])
    snap = Snap(["demix_filter", "output"])
    m = AuxIVA(spatial_algorithm=algo, contrast_fn=lambda y: np.linalg.norm(y, axis=1) ** p,
               d_contrast_fn=lambda r: p * r ** (p - 1), callbacks=snap)
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]))
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    _compare_snapshots(g, snap)
    assert rel_err(Y, g["final_output"]) < TOL
    # batched: the closures see one mixture at a time
    mb = AuxIVA(spatial_algorithm=algo, contrast_fn=lambda y: np.linalg.norm(y, axis=1) ** p,
                d_contrast_fn=lambda r: p * r ** (p - 1))
    Yb = mb(np.stack([g["X"], 2 * g["X"]]), n_iter=int(g["meta_n_iter"]))
    assert rel_err(Yb[0], g["final_output"]) < TOL
    np.testing.assert_allclose(np.asarray(mb.loss)[:, 0], g["loss"], rtol=LOSS_RTOL)


def test_all_channel_scale_restoration_against_golden():
    """projection_back / minimal_distortion_principle with reference_id=None: every channel in turn,
    stacked on a new leading axis (ref: projection_back.py:92-95, :113-116; mdp :34-35)."""
    from ssspy_amd.algorithm import minimal_distortion_principle, projection_back

    g = load_golden("restoration_all_channels")
    for N in (2, 3, 4):
        X, Y, W = (g["n{}_{}".format(N, k)] for k in "XYW")
        out = projection_back(W, reference_id=None)
        assert out.shape == g["n{}_pb_filter".format(N)].shape
        assert rel_err(out, g["n{}_pb_filter".format(N)]) < 1e-11
        out = projection_back(Y, reference=X, reference_id=None)
        assert out.shape == (N, N) + Y.shape[1:]
        assert rel_err(out, g["n{}_pb_output".format(N)]) < 1e-11
        out = minimal_distortion_principle(Y, reference=X, reference_id=None)
        assert rel_err(out, g["n{}_mdp_output".format(N)]) < 1e-11


def test_iss_frame_power_cache_follows_output_writes():
    """AuxIVA in the ISS state keeps the frame powers the fused sweep leaves behind; any later write
    to ``output`` (minimal-distortion scaling at the end of a call, an assignment from user code)
    must retire them: continuing to iterate afterwards equals the oracle doing the same."""
    from oracle import spatial as sp
    from oracle.iva import AuxIVAOracle
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(77, 3, 24, 40)
    m = AuxLaplaceIVA(spatial_algorithm="ISS", scale_restoration="minimal_distortion_principle")
    m(X, n_iter=3)
    m.update_once()  # weights must come from the rescaled output, not the pre-MDP frame powers
    ref = AuxIVAOracle(spatial_algorithm="ISS", contrast="laplace",
                       scale_restoration="minimal_distortion_principle")
    ref.run(X, n_iter=3)
    ref.update_once()
    assert rel_err(m.output, ref.output) < TOL
    assert m.compute_loss() == pytest.approx(ref.compute_loss(), rel=LOSS_RTOL)
    m.output = 3.0 * ref.output  # assignment from user code
    ref.output = 3.0 * ref.output
    m.update_once()
    ref.update_once()
    assert rel_err(m.output, ref.output) < TOL
    assert sp is not None


def test_state_snapshots_are_read_only_and_assignment_uploads():
    """Attributes read during the iteration are snapshots of HBM buffers: an in-place edit would be
    lost, so it fails loudly; assigning the attribute takes effect.  The array ``__call__`` returns
    is writable like the reference's."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(5, 2, 16, 24)
    seen = {}

    def cb(method):
        seen["flag"] = method.basis.flags.writeable
        with pytest.raises(ValueError):
            method.demix_filter[...] = 0

    m = GaussILRMA(n_basis=3, callbacks=cb, rng=np.random.default_rng(0))
    Y = m(X, n_iter=1)
    assert seen["flag"] is False
    Y *= 1.0  # returned estimate is writable
    new_basis = np.full_like(m.basis, 0.5)
    m.basis = new_basis
    m.update_activation_mm()
    from oracle.ilrma import GaussILRMAOracle
    assert np.array_equal(m.basis, new_basis)
    assert GaussILRMAOracle is not None


def test_mnmf_separate_rejects_other_shapes():
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(9, 2, 12, 20)
    for cls in (FastGaussMNMF, GaussMNMF):
        m = cls(n_basis=2, rng=np.random.default_rng(0))
        m(X, n_iter=1)
        assert m.separate(X).shape == X.shape
        with pytest.raises(ValueError):
            m.separate(np.concatenate([X, X], axis=-1))  # more frames than the activation has
        with pytest.raises(ValueError):
            m.separate(np.stack([X, X]))  # a batch against single-mixture parameters


def test_solve_keeps_real_systems_real():
    from ssspy_amd.linalg import solve

    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 4, 4)) + 4 * np.eye(4)
    b = rng.standard_normal((5, 4))
    x = solve(a, b)
    assert x.dtype == np.float64
    np.testing.assert_allclose(x, np.linalg.solve(a, b[..., None])[..., 0], rtol=1e-10)
    assert solve(a.astype(complex), b).dtype == np.complex128


# ------------------------------------------------------------------------------- deferred loss (round 2)
@pytest.mark.parametrize("case", ["gilrma_ip1_n2", "gilrma_ip1_n3", "gilrma_ip1_n4", "gilrma_ip1_n4_p1",
                                  "gilrma_ip1_n2_add", "gilrma_ip1_n3_raw", "tilrma_ip1_n3",
                                  "ggdilrma_ip1_n3", "gilrma_me_ip1_n3"])
def test_deferred_loss_loop_against_golden(case):
    """Without callbacks the separator takes the loss of iteration t from the basis pass of iteration
    t + 1 (one fused C-ABI call per iteration, a dedicated loss pass only at the end): the loss list,
    filters and output must equal the reference's record_loss=True run.
    ref: ssspy/bss/base.py:68-77, ssspy/bss/ilrma.py:1910-1967."""
    g = load_golden(case)
    model = (str(g["meta_model"]), float(g["meta_model_param"])) if "meta_model" in g else ("gauss", None)
    cls = _ilrma_class(model)
    kw = dict(n_basis=int(g["meta_n_basis"]), spatial_algorithm=str(g["meta_algo"]),
              domain=float(g["meta_domain"]), flooring_fn=_flooring_fn(g),
              normalization=_option(g["meta_normalization"]),
              scale_restoration=_option(g["meta_scale_restoration"]))
    if "meta_source_algorithm" in g:
        kw["source_algorithm"] = str(g["meta_source_algorithm"])
    if model[0] == "t":
        kw["dof"] = model[1]
    elif model[0] == "ggd":
        kw["beta"] = model[1]
    m = cls(**kw)
    assert m._iterate_with_deferred_loss.__func__ is not None
    Y = m(g["X"], n_iter=int(g["meta_n_iter"]), basis=g["basis0"], activation=g["activation0"])
    assert len(m.loss) == int(g["meta_n_iter"]) + 1 and all(type(v) is float for v in m.loss)
    np.testing.assert_allclose(m.loss, g["loss"], rtol=LOSS_RTOL)
    assert rel_err(Y, g["final_output"]) < TOL
    # no initial call: the reference records only the losses after the iterations
    m2 = cls(**kw)
    m2(g["X"], n_iter=int(g["meta_n_iter"]), initial_call=False, basis=g["basis0"],
       activation=g["activation0"])
    np.testing.assert_allclose(m2.loss, g["loss"][1:], rtol=LOSS_RTOL)


def test_deferred_loss_batched_and_large_batch_path():
    """The by-product in the unsplit (whole rounds) and split (tail) blocks of the batched launch:
    600 tiny mixtures, loss lists of the first / middle / last against the oracle."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, N, F, T, K = 600, 3, 18, 33, 4
    rng = np.random.default_rng(6)
    X = np.stack([nmf_mixture(8000 + b, N, F, T) for b in range(B)])
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    m = GaussILRMA(n_basis=K)
    m(X, n_iter=3, basis=basis, activation=act)
    loss = np.asarray(m.loss)
    assert loss.shape == (4, B)
    for b in (0, 299, B - 1):
        ref = GaussILRMAOracle(n_basis=K)
        ref.run(X[b], n_iter=3, basis=basis[b], activation=act[b])
        np.testing.assert_allclose(loss[:, b], ref.loss, rtol=LOSS_RTOL)


# ------------------------------------------------------------------------------- FastMNMF, general shapes
@pytest.mark.parametrize("M,N,algo", [(5, 5, "IP"), (6, 2, "IP"), (8, 8, "IP"), (3, 1, "IP"),
                                      (6, 3, "IP2"), (4, 6, "IP")])
def test_fast_gauss_mnmf_general_shapes_against_oracle(M, N, algo):
    """More than 4 channels or sources (and a single source): the point-wise general path of
    fmnmf_generic.hip -- every parameter, the loss list and the Wiener-filter output against the
    oracle; batch of two equals the single runs; the step methods equal the fused update.
    ref: ssspy/bss/mnmf.py:1278-1303, :1174-1217."""
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    F, T, K = 9, 70, 3
    rng = np.random.default_rng(100 * M + N)
    Xs = np.stack([nmf_mixture(900 + M + b, M, F, T) for b in range(2)])
    kw = dict(basis=rng.random((2, N, F, K)), activation=rng.random((2, N, K, T)),
              spatial=rng.random((2, F, N, M)))
    mb = FastGaussMNMF(n_basis=K, n_sources=N, diagonalizer_algorithm=algo)
    Yb = mb(Xs, n_iter=4, **kw)
    for b in range(2):
        one = {k: v[b] for k, v in kw.items()}
        if algo == "IP":
            ref = FastGaussMNMFOracle(n_basis=K, n_sources=N)
            Yr = ref.run(Xs[b], n_iter=4, **{k: v.copy() for k, v in one.items()})
            np.testing.assert_allclose(np.asarray(mb.loss)[:, b], ref.loss, rtol=1e-8)
            for name in ("diagonalizer", "spatial", "basis", "activation"):
                assert rel_err(getattr(mb, name)[b], getattr(ref, name)) < 1e-7, name
            assert rel_err(Yb[b], Yr) < 1e-6
        m1 = FastGaussMNMF(n_basis=K, n_sources=N, diagonalizer_algorithm=algo)
        Y1 = m1(Xs[b], n_iter=4, **one)
        assert rel_err(Yb[b], Y1) < 1e-11
        np.testing.assert_allclose(np.asarray(mb.loss)[:, b], m1.loss, rtol=1e-11)
    # step methods one by one == the fused update
    ms = FastGaussMNMF(n_basis=K, n_sources=N, diagonalizer_algorithm=algo)
    ms._bind_input(Xs[0])
    ms._reset(**{k: v[0] for k, v in kw.items()})
    mf = FastGaussMNMF(n_basis=K, n_sources=N, diagonalizer_algorithm=algo)
    mf._bind_input(Xs[0])
    mf._reset(**{k: v[0] for k, v in kw.items()})
    mf.update_once()
    ms.update_basis()
    ms.update_activation()
    ms.update_diagonalizer()
    ms.update_spatial()
    ms.normalize()
    for name in ("diagonalizer", "spatial", "basis", "activation"):
        assert rel_err(getattr(ms, name), getattr(mf, name)) < 1e-12, name


def test_large_n_basis_partitioned_and_mnmf():
    """n_basis above 64 (the former limit of the partitioning kernels' LDS tables): partitioned
    GaussILRMA, FastGaussMNMF and GaussMNMF with 100 bases against the oracle."""
    from oracle.gmnmf import GaussMNMFOracle
    from oracle.ilrma import GaussILRMAOracle
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 3, 10, 40, 100
    X = nmf_mixture(31, N, F, T)
    rng = np.random.default_rng(7)
    basis, act = rng.random((F, K)), rng.random((K, T))
    latent = rng.random((N, K))
    latent /= latent.sum(axis=0)
    m = GaussILRMA(n_basis=K, partitioning=True)
    Y = m(X, n_iter=3, basis=basis, activation=act, latent=latent)
    ref = GaussILRMAOracle(n_basis=K, partitioning=True)
    Yr = ref.run(X, n_iter=3, basis=basis, activation=act, latent=latent)
    assert rel_err(Y, Yr) < TOL
    np.testing.assert_allclose(m.loss, ref.loss, rtol=LOSS_RTOL)
    b3, a3 = rng.random((N, F, K)), rng.random((N, K, T))
    sp0 = rng.random((F, N, N))
    mf = FastGaussMNMF(n_basis=K)
    Yf = mf(X, n_iter=2, basis=b3, activation=a3, spatial=sp0)
    rf = FastGaussMNMFOracle(n_basis=K)
    Yfr = rf.run(X, n_iter=2, basis=b3, activation=a3, spatial=sp0.copy())
    np.testing.assert_allclose(mf.loss, rf.loss, rtol=1e-8)
    assert rel_err(Yf, Yfr) < 1e-6
    mg = GaussMNMF(n_basis=K)
    Yg = mg(X, n_iter=2, basis=b3, activation=a3)
    rg = GaussMNMFOracle(n_basis=K)
    Ygr = rg.run(X, n_iter=2, basis=b3, activation=a3)
    np.testing.assert_allclose(mg.loss, rg.loss, rtol=1e-7)
    assert rel_err(Yg, Ygr) < 1e-6


@pytest.mark.parametrize("model,domain,algo,src", [(("gauss", None), 2, "IP", "MM"), (("gauss", None), 1, "IP", "MM"),
                                                   (("t", 4.0), 2, "IP", "MM"), (("ggd", 1.3), 2, "IP", "MM"),
                                                   (("gauss", None), 2, "ISS", "MM"), (("gauss", None), 2, "IP", "ME"),
                                                   (("t", 4.0), 2, "IP2", "MM")])
@pytest.mark.parametrize("K,T", [(20, 50), (32, 33)])
def test_ilrma_wide_basis_tuned_path_against_oracle(model, domain, algo, src, K, T):
    """16 < n_basis <= 32 on the tuned kernels (two k-tile work items per bin group; basis written out
    of place): every source model, filter and ISS states, odd frame counts, a batch that mixes whole
    rounds and split tail items."""
    from oracle.ilrma import GaussILRMAOracle

    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, B = 3, 70, 2
    rng = np.random.default_rng(K + T)
    X = np.stack([nmf_mixture(500 + b, N, F, T) for b in range(B)])
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    cls = _ilrma_class(model)
    kw = dict(n_basis=K, spatial_algorithm=algo, domain=domain, source_algorithm=src)
    if model[0] == "t":
        kw["dof"] = model[1]
    elif model[0] == "ggd":
        kw["beta"] = model[1]
    m = cls(**kw)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    for b in range(B):
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, domain=domain, model=model,
                               source_algorithm=src)
        Yr = ref.run(X[b], n_iter=3, basis=basis[b], activation=act[b])
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref.loss, rtol=LOSS_RTOL)
        assert rel_err(m.basis[b], ref.basis) < TOL and rel_err(m.activation[b], ref.activation) < TOL
        if algo == "IP2":
            assert rel_err(Y[b], Yr) < 1e-7
        else:
            assert rel_err(Y[b], Yr) < TOL


def test_ilrma_wide_basis_large_batch():
    """n_basis = 24 with 300 tiny mixtures: more work items than resident workgroups, so the launch
    has whole rounds and a split tail; first / middle / last mixture against the oracle."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    B, N, F, T, K = 300, 4, 70, 40, 24
    rng = np.random.default_rng(9)
    X = np.stack([nmf_mixture(9000 + b, N, F, T) for b in range(B)])
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    m = GaussILRMA(n_basis=K)
    Y = m(X, n_iter=2, basis=basis, activation=act)
    for b in (0, 150, B - 1):
        ref = GaussILRMAOracle(n_basis=K)
        Yr = ref.run(X[b], n_iter=2, basis=basis[b], activation=act[b])
        assert rel_err(Y[b], Yr) < TOL
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref.loss, rtol=LOSS_RTOL)


@pytest.mark.parametrize("model,N,algo,K", [(("gauss", None), 6, "IP", 5), (("gauss", None), 8, "IP", 16),
                                            (("gauss", None), 8, "ISS", 7), (("t", 5.0), 8, "IP", 4),
                                            (("ggd", 1.4), 6, "IP2", 6), (("gauss", None), 8, "IP", 24)])
def test_ilrma_wide_mixture_grouped_sources_against_oracle(model, N, algo, K):
    """6 or 8 sources: the NMF passes walk the separated spectrogram as N / G groups of G <= 4 sources
    through the tuned kernels (y = W x formed once per iteration).  Batch of two against the oracle;
    the step methods against the fused update."""
    from oracle.ilrma import GaussILRMAOracle

    from ssspy_amd.utils.dataset import nmf_mixture

    F, T, B = 40, 60, 2
    rng = np.random.default_rng(N * 10 + K)
    X = np.stack([nmf_mixture(600 + b, N, F, T) for b in range(B)])
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    cls = _ilrma_class(model)
    kw = dict(n_basis=K, spatial_algorithm=algo)
    if model[0] == "t":
        kw["dof"] = model[1]
    elif model[0] == "ggd":
        kw["beta"] = model[1]
    m = cls(**kw)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    for b in range(B):
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, model=model)
        Yr = ref.run(X[b], n_iter=3, basis=basis[b], activation=act[b])
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref.loss, rtol=LOSS_RTOL)
        assert rel_err(m.basis[b], ref.basis) < TOL and rel_err(m.activation[b], ref.activation) < TOL
        assert rel_err(Y[b], Yr) < (1e-7 if algo == "IP2" else TOL)
    if algo == "IP":
        ms = cls(**kw)
        ms._bind_input(X)
        ms._reset(flooring_fn=ms.flooring_fn, basis=basis, activation=act)
        mf = cls(**kw)
        mf._bind_input(X)
        mf._reset(flooring_fn=mf.flooring_fn, basis=basis, activation=act)
        mf.update_once()
        ms.update_source_model()
        ms.update_spatial_model()
        ms.normalize()
        # (the step method and the fused update may take different covariance kernels for a wide
        # heavy-tailed mixture: equal up to rounding, not bit for bit)
        for name in ("demix_filter", "basis", "activation"):
            assert rel_err(getattr(ms, name), getattr(mf, name)) < 1e-10, name


@pytest.mark.parametrize("model,N,B,algo,K", [
    (("gauss", None), 5, 1, "IP", 4),     # 5 sources: one group of 3 and one of 2
    (("gauss", None), 5, 2, "IP", 16),    # 10: two groups of 4, one of 2
    (("gauss", None), 7, 1, "ISS", 6),    # 7: 4 + 3
    (("t", 5.0), 7, 3, "IP", 3),          # 21: four groups of 4, then 3 + 2
    (("ggd", 1.4), 5, 3, "IP2", 5),       # 15: three groups of 4, one of 3
    (("gauss", None), 7, 2, "IP", 24),    # 14 with two k-tile work items per bin group
])
def test_ilrma_five_and_seven_sources_against_oracle(model, N, B, algo, K):
    """Source counts without a divisor in {2, 3, 4}: the B N sources are cut into runs of groups of
    4 with closing groups of 3 / 2, one tuned launch per run (ilrma_api.hip: source_runs)."""
    from oracle.ilrma import GaussILRMAOracle

    from ssspy_amd.utils.dataset import nmf_mixture

    F, T = 40, 60
    rng = np.random.default_rng(N * 10 + K + B)
    X = np.stack([nmf_mixture(700 + b, N, F, T) for b in range(B)])
    basis, act = rng.random((B, N, F, K)), rng.random((B, N, K, T))
    cls = _ilrma_class(model)
    kw = dict(n_basis=K, spatial_algorithm=algo)
    if model[0] == "t":
        kw["dof"] = model[1]
    elif model[0] == "ggd":
        kw["beta"] = model[1]
    m = cls(**kw)
    Y = m(X, n_iter=3, basis=basis, activation=act)
    for b in range(B):
        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo, model=model)
        Yr = ref.run(X[b], n_iter=3, basis=basis[b], activation=act[b])
        # (IP2: eigenvectors of nearly degenerate pairs; the loss is a difference of large terms)
        np.testing.assert_allclose(np.asarray(m.loss)[:, b], ref.loss,
                                   rtol=1e-6 if algo == "IP2" else LOSS_RTOL)
        # (IP2 on GGD sources: near-degenerate eigen pairs amplify rounding; north_star asks 1e-4)
        tol = 1e-6 if algo == "IP2" else TOL
        assert rel_err(m.basis[b], ref.basis) < tol and rel_err(m.activation[b], ref.activation) < tol
        assert rel_err(Y[b], Yr) < tol


# ------------------------------------------------------------------------------- rng-drawn state
class _Snap0:
    """Callback that keeps the state at the initial call, after iteration 1 and after the last."""

    def __init__(self, names, n_iter):
        self.names, self.n_iter, self.count, self.store = names, n_iter, -1, {}

    def __call__(self, m):
        self.count += 1
        if self.count in (0, 1, self.n_iter):
            for name in self.names:
                v = getattr(m, name, None)
                if v is not None:
                    self.store["it{}_{}".format(self.count, name)] = np.array(v, copy=True)


def _separator_for(g, snap, flooring_fn="default"):
    """Product separator for a ``rnginit_*`` / ``customfloor_*`` fixture, seeded like the reference
    run that made it; nothing is injected."""
    from ssspy_amd.bss.ilrma import TILRMA, GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF

    kind = str(g["meta_kind"]).split("_")[-1]
    rng = np.random.default_rng(int(g["meta_seed"]) + 3)
    K = int(g["meta_n_basis"])
    kw = {} if type(flooring_fn) is str else {"flooring_fn": flooring_fn}
    part = bool(g["meta_partitioning"]) if "meta_partitioning" in g else False
    if kind == "ilrma":
        common = dict(n_basis=K, spatial_algorithm=str(g["meta_spatial_algorithm"]),
                      partitioning=part, callbacks=snap, rng=rng, **kw)
        if "meta_model" in g and str(g["meta_model"]) == "t":
            return TILRMA(dof=float(g["meta_model_param"]), **common)
        if "meta_model" in g and str(g["meta_model"]) == "ggd":
            from ssspy_amd.bss.ilrma import GGDILRMA

            return GGDILRMA(beta=float(g["meta_model_param"]), **common)
        return GaussILRMA(**common)
    if kind == "iva":
        return AuxLaplaceIVA(spatial_algorithm=str(g["meta_spatial_algorithm"]), callbacks=snap, **kw)
    if kind == "gaussiva":
        from ssspy_amd.bss.iva import AuxGaussIVA

        return AuxGaussIVA(spatial_algorithm=str(g["meta_spatial_algorithm"]), callbacks=snap, **kw)
    if kind == "fmnmf":
        return FastGaussMNMF(n_basis=K, callbacks=snap, rng=rng, **kw)
    return GaussMNMF(n_basis=K, partitioning=part, callbacks=snap, rng=rng, **kw)


_STATE_NAMES = ["latent", "basis", "activation", "demix_filter", "diagonalizer", "spatial", "output",
                "variance"]


def _replay_uninjected(g, flooring_fn="default", tol=TOL):
    n_iter = int(g["meta_n_iter"])
    snap = _Snap0(_STATE_NAMES, n_iter)
    m = _separator_for(g, snap, flooring_fn)
    Y = m(g["X"], n_iter=n_iter)
    checked = 0
    for key, value in snap.store.items():
        if key not in g:
            continue
        name = key.split("_", 1)[1]
        pairwise = "meta_spatial_algorithm" in g and str(g["meta_spatial_algorithm"]) in ("IP2", "ISS2")
        if key.startswith("it0_") and name not in ("output", "variance"):
            np.testing.assert_array_equal(value, g[key], err_msg=key)  # the draws themselves
        elif (pairwise and not key.startswith("it0_") and name in ("demix_filter", "output")
              and g["X"].shape[0] == 2):
            # (two sources: the second visit of the pair has a noise-derived phase, see above)
            assert rel_err_up_to_phase(value, g[key], name) < tol, key
        else:
            assert rel_err(value, g[key]) < tol, key
        checked += 1
    assert checked >= 3  # (the ISS state keeps no filters: output at three points)
    np.testing.assert_allclose(m.loss, g["loss"], rtol=max(LOSS_RTOL, tol * 0.1))
    assert rel_err(Y, g["final_output"]) < tol


@pytest.mark.parametrize("case", ["rnginit_gilrma_n3", "rnginit_gilrma_part_n3",
                                  "rnginit_gilrma_part_iss_n2", "rnginit_tilrma_part_n2",
                                  "rnginit_fmnmf_m3", "rnginit_gmnmf_m2", "rnginit_gmnmf_part_m2"])
def test_rng_drawn_initial_state_against_golden(case):
    """Only ``rng=default_rng(s)`` is given: the separator must draw its parameters in the
    reference's order (ssspy/bss/ilrma.py:230-266, mnmf.py:221-254, :535-538, :595) and then
    follow the reference run."""
    _replay_uninjected(load_golden(case), tol=1e-7 if "gmnmf" in case else TOL)


# ------------------------------------------------------------------------------- determinism
@pytest.mark.parametrize("kind", ["auxiva_ip", "auxiva_iss", "ilrma_iss", "ilrma_ip", "gmnmf",
                                  "fmnmf", "fmnmf_wide"])
def test_state_is_bitwise_reproducible(kind):
    """No fp64 atomics on anything the state depends on (frame powers of AuxIVA and of the fused ISS
    sweep, output power of the ISS normalisation, GaussMNMF's activation sums) nor on the ILRMA /
    AuxIVA loss terms: two runs from the same input give the same bits, for a single mixture and
    for a batch."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = (2, 24, 40, 3) if kind == "gmnmf" else (4, 257, 300, 6)
    if kind == "fmnmf_wide":  # the point-wise path above 4 channels
        N, F, T, K = 5, 33, 64, 3
    for B in (1, 5):
        X = np.stack([nmf_mixture(950 + b, N, F, T) for b in range(B)])

        def run():
            if kind.startswith("auxiva"):
                m = AuxLaplaceIVA(spatial_algorithm="IP" if kind.endswith("ip") else "ISS",
                                  record_loss=True)
                return m(X, n_iter=6), np.asarray(m.loss)
            if kind == "gmnmf":
                m = GaussMNMF(n_basis=K, record_loss=True, rng=np.random.default_rng(3))
                return m(X, n_iter=3), np.asarray(m.loss)
            if kind.startswith("fmnmf"):
                m = FastGaussMNMF(n_basis=K, record_loss=True, rng=np.random.default_rng(3))
                return m(X, n_iter=4), np.asarray(m.loss)
            m = GaussILRMA(n_basis=K, spatial_algorithm="ISS" if kind.endswith("iss") else "IP",
                           record_loss=True, rng=np.random.default_rng(3))
            return m(X, n_iter=6), np.asarray(m.loss)

        (Y1, L1), (Y2, L2) = run(), run()
        assert np.array_equal(Y1, Y2), (kind, B)
        # the recorded losses too (per-wave / per-block shares folded in a fixed order: ILRMA both as
        # the by-product of the basis pass and from the loss pass, ISS with the tracked
        # log-determinant, FastMNMF from the hand-over and from x, GaussMNMF)
        assert np.array_equal(L1, L2), (kind, B)


# ------------------------------------------------------------------------------- arbitrary floors
def _golden_custom_floor(x):
    """Same function as tests/golden/make_golden.py:custom_floor (a fixture cannot carry code)."""
    return np.maximum(x, 1e-8) + 1e-12


@pytest.mark.parametrize("case", ["customfloor_gilrma_ip1_n3", "customfloor_gilrma_iss1_n2",
                                  "customfloor_auxlap_ip1_n3", "customfloor_auxlap_iss1_n2",
                                  "customfloor_gilrma_ip2_n3", "customfloor_gilrma_iss2_n4",
                                  "customfloor_gilrma_part_ip1_n3", "customfloor_gilrma_part_iss1_n2",
                                  "customfloor_auxlap_ip2_n3", "customfloor_auxlap_iss2_n3",
                                  "customfloor_auxgauss_ip1_n3", "customfloor_auxgauss_iss1_n2",
                                  "customfloor_auxgauss_ip2_n3", "customfloor_fmnmf_m3",
                                  "customfloor_tilrma_ip2_n3", "customfloor_tilrma_iss2_n3",
                                  "customfloor_ggdilrma_ip1_n3", "customfloor_ggdilrma_iss1_n2",
                                  "customfloor_ggdilrma_ip2_n3", "customfloor_ggdilrma_iss2_n3"])
def test_arbitrary_flooring_callable_against_golden(case):
    """``flooring_fn`` may be any callable in the reference (ssspy/bss/ilrma.py:70-89).  One that is
    none of the three built-in floors is evaluated on the host on the small arrays it acts on (basis,
    activation, IP1 / ISS1 denominators, normalisation scales, AuxIVA's frame norms); the passes over
    the spectrograms stay on the device."""
    _replay_uninjected(load_golden(case), flooring_fn=_golden_custom_floor)


@pytest.mark.parametrize("kind", ["ilrma_ip", "ilrma_iss", "tilrma_ip", "tilrma_ip2", "tilrma_iss2",
                                  "auxiva_ip", "auxiva_iss"])
def test_host_evaluated_floor_equals_the_kernel_floor(kind):
    """A callable the kernels do not recognise but that computes max(x, eps) takes the host path
    (split steps, floor on the small arrays); the recognised ``max_flooring`` runs inside the
    kernels.  Same mathematics, different code on both sides of the boundary: the results agree
    to rounding, on a batch and at a size the goldens do not reach."""
    import functools

    from ssspy_amd.bss.ilrma import TILRMA, GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.special.flooring import max_flooring
    from ssspy_amd.utils.dataset import nmf_mixture

    eps = 1e-9
    B, N, F, T = 3, 4, 129, 150
    X = np.stack([nmf_mixture(620 + b, N, F, T) for b in range(B)])

    def make(floor):
        rng = np.random.default_rng(8)
        if kind == "ilrma_ip":
            return GaussILRMA(n_basis=6, flooring_fn=floor, rng=rng)
        if kind == "ilrma_iss":
            return GaussILRMA(n_basis=6, spatial_algorithm="ISS", flooring_fn=floor, rng=rng)
        if kind == "tilrma_ip":
            return TILRMA(n_basis=6, dof=4.0, flooring_fn=floor, rng=rng)
        if kind in ("tilrma_ip2", "tilrma_iss2"):  # (round 6: the t weights hold no floor)
            return TILRMA(n_basis=6, dof=4.0, spatial_algorithm=kind[7:].upper(), flooring_fn=floor,
                          rng=rng)
        return AuxLaplaceIVA(spatial_algorithm="IP" if kind.endswith("ip") else "ISS",
                             flooring_fn=floor)

    dev = make(functools.partial(max_flooring, eps=eps))
    host = make(lambda x: np.maximum(x, eps))
    Yd, Yh = dev(X, n_iter=5), host(X, n_iter=5)
    assert rel_err(Yh, Yd) < 1e-10
    np.testing.assert_allclose(np.asarray(host.loss), np.asarray(dev.loss), rtol=1e-10)


def test_arbitrary_flooring_callable_unsupported_paths_fail_loudly():
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(5, 3, 12, 24)
    with pytest.raises(NotImplementedError, match="IPA"):
        GaussILRMA(n_basis=2, spatial_algorithm="IPA", flooring_fn=_golden_custom_floor)(X, n_iter=1)
    from ssspy_amd.bss.ilrma import TILRMA

    from ssspy_amd.bss.ilrma import GGDILRMA

    # (the t model's weights hold no floor: TILRMA takes any callable with IP2 / ISS2 since round 6;
    #  GGD's floor |y|^(2 - beta) per element of the spectrogram)
    # (round 6: TILRMA and GGDILRMA take any callable -- the t model's weights hold no floor, GGD's
    #  floor on |y|^(2 - beta) is evaluated on the host; partitioning with them does not)
    TILRMA(n_basis=2, dof=4.0, spatial_algorithm="IP2", flooring_fn=_golden_custom_floor)(X, n_iter=1)
    GGDILRMA(n_basis=2, beta=1.0, spatial_algorithm="IP2", flooring_fn=_golden_custom_floor)(X, n_iter=1)
    with pytest.raises(NotImplementedError, match="[Pp]artitioning"):
        GGDILRMA(n_basis=2, beta=1.0, partitioning=True,
                 flooring_fn=_golden_custom_floor)(X, n_iter=1)
    from ssspy_amd.bss.mnmf import GaussMNMF

    with pytest.raises(NotImplementedError, match="GaussMNMF"):
        GaussMNMF(n_basis=2, flooring_fn=_golden_custom_floor)(X, n_iter=1)
    assert FastGaussMNMF is not None  # (FastGaussMNMF takes any callable since round 5: see below)


@pytest.mark.parametrize("algo,N,M,B", [("IP", 3, 3, 1), ("IP", 4, 4, 3), ("IP2", 3, 4, 2), ("IP", 2, 6, 1)])
def test_fast_gauss_mnmf_host_evaluated_floor_equals_the_kernel_floor(algo, N, M, B, monkeypatch):
    """Round 5: FastGaussMNMF with a flooring callable the kernels do not recognise -- the steps run
    one by one with the floor off, the callable on basis / activation / IP denominators / psi on the
    host, and the Wiener filter is split at its eigenvalue floor (stage 1: eigen-decomposition of
    every R_ij to HBM; the callable on the (F, T, M) eigenvalues; stage 2).  With max(x, eps) as the
    callable the result must equal the recognised max_flooring inside the kernels, with an eps large
    enough for the Wiener floor to act (1e-2) and on the point-wise path above 4 channels."""
    import functools

    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.special.flooring import max_flooring
    from ssspy_amd.utils.dataset import nmf_mixture

    from ssspy_amd import _ops

    F, T, K = 33, 48, 3
    X = np.stack([nmf_mixture(700 + b, M, F, T) for b in range(B)])
    if B == 1:
        X = X[0]
    # (round 6: the split Wiener filter walks the batch in chunks that fit a byte budget for the
    #  eigenvectors; with room for two mixtures the batch of three takes a chunk of 2 and one of 1)
    monkeypatch.setattr(_ops, "_HOST_FLOOR_EIG_BYTES", 2 * F * T * M * M * 16)
    for eps in (1e-10, 1e-2):
        def make(floor):
            return FastGaussMNMF(n_basis=K, n_sources=N, diagonalizer_algorithm=algo,
                                 flooring_fn=floor, rng=np.random.default_rng(5))

        dev = make(functools.partial(max_flooring, eps=eps))
        host = make(lambda x, e=eps: np.maximum(x, e))
        Yd, Yh = dev(X, n_iter=4), host(X, n_iter=4)
        assert rel_err(Yh, Yd) < 1e-8
        np.testing.assert_allclose(np.asarray(host.loss), np.asarray(dev.loss), rtol=1e-9)
        assert rel_err(host.basis, dev.basis) < 1e-9
        assert rel_err(host.spatial, dev.spatial) < 1e-9


# ------------------------------------------------------------------------------- non-finite inputs
def _same_nonfinite(out, ref, tol=1e-9):
    """The reference propagates NaN / Inf silently (e.g. num / denom with a zero denominator,
    ssspy/bss/ilrma.py:1125): the same elements must be non-finite, the finite ones equal."""
    fo, fr = np.isfinite(out), np.isfinite(ref)
    assert np.array_equal(fo, fr), "non-finite pattern differs: {} vs {} elements".format(
        int((~fo).sum()), int((~fr).sum()))
    assert (~fr).any() and fr.any()
    assert rel_err(out[fr], ref[fr]) < tol


@pytest.mark.parametrize("N,T", [(2, 70), (4, 300), (8, 1000), (8, 257)])
def test_fused_iss_propagates_non_finite_samples_like_the_reference(N, T):
    """NaN / Inf samples in a few bins, frame counts that leave padding lanes in the register slab
    (T not a multiple of 256; round 2's advisor found a 0 * Inf there): exactly the bins the
    reference turns non-finite are non-finite, every other bin is untouched."""
    from oracle import spatial as sp
    from ssspy_amd.bss._update_spatial_model import update_by_iss1

    rng = np.random.default_rng(N * 1000 + T)
    F = 9
    Y = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    varphi = 1 / (rng.random((N, 1, T)) + 0.1)
    Y[0, 1, T - 1] = np.inf          # the last valid frame: next to the padding
    Y[N - 1, 4, 0] = np.nan
    Y[1 % N, 7, T // 2] = -np.inf + 1j
    with np.errstate(all="ignore"):
        ref = sp.update_by_iss1(Y, varphi)
    out = update_by_iss1(Y, varphi)
    _same_nonfinite(out, ref)
    bad_bins = sorted(set(np.where(~np.isfinite(ref))[1].tolist()))
    assert bad_bins == [1, 4, 7]


@pytest.mark.parametrize("N", [6, 8])
def test_wide_covariance_propagates_non_finite_samples_like_the_reference(N):
    """The matrix-core covariance of 6..8 channels (wide_cov.hip): frames beyond T are clamped loads
    with weight 0; a non-finite sample must poison its own bin only."""
    from ssspy_amd import _device as dv
    from ssspy_amd import _lib, _ops

    rng = np.random.default_rng(N)
    F, T = 10, 77  # 77 = 9 slabs of 8 frames + 5: padding lanes in the last slab
    X = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    w = 1 / (rng.random((N, F, T)) + 0.1)
    X[2, 3, T - 1] = np.inf
    X[0, 8, 5] = np.nan
    with np.errstate(all="ignore"):
        XX = X[:, None] * X[None].conj()                       # (a, c, F, T)
        ref = np.mean(w[:, None, None] * XX[None], axis=-1).transpose(3, 0, 1, 2)  # (F, s, a, c)
    U = dv.to_host(_ops.weighted_covariance(dv.to_device(X[None]), dv.to_device(w[None]),
                                            _lib.WEIGHT_BIN_FRAME, N))[0]
    fo, fr = np.isfinite(U), np.isfinite(ref)
    bad_out = sorted(set(np.where(~fo)[0].tolist()))
    assert bad_out == sorted(set(np.where(~fr)[0].tolist())) == [3, 8]
    good = [i for i in range(F) if i not in (3, 8)]
    assert rel_err(U[good], ref[good]) < 1e-11


def test_zero_denominators_propagate_like_the_reference():
    """An all-zero activation row makes num = den = 0 in the basis update: the reference computes
    0 / 0 = NaN without a word (ssspy/bss/ilrma.py:1125, flooring_fn=None keeps it); so does the
    device pass, in the same elements."""
    from oracle.ilrma import GaussILRMAOracle
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K = 3, 20, 40, 4
    X = nmf_mixture(77, N, F, T)
    basis = np.random.default_rng(1).random((N, F, K))
    act = np.random.default_rng(2).random((N, K, T))
    act[1, 2, :] = 0.0
    ref = GaussILRMAOracle(n_basis=K, flooring=("none", 0.0), normalization=False, record_loss=False)
    ref.reset(X, basis=basis, activation=act)
    with np.errstate(all="ignore"):
        ref.update_basis()
    m = GaussILRMA(n_basis=K, flooring_fn=None, normalization=False, record_loss=False)
    m._bind_input(X)
    m._reset(flooring_fn=m.flooring_fn, basis=basis, activation=act)
    m.update_basis_mm()
    out = np.asarray(m.basis)
    _same_nonfinite(out, ref.basis, tol=1e-11)
    assert np.isnan(out[1, :, 2]).all() and np.isfinite(out[0]).all()


@pytest.mark.parametrize("cls,K", [("fast", 17), ("fast", 300), ("fast", 1024), ("gauss", 300),
                                   ("gauss_part", 520)])
def test_mnmf_n_basis_beyond_256_against_oracle(cls, K):
    """The reference puts no bound on n_basis (ssspy/bss/mnmf.py:1112-1153, :681-763); the MNMF entry
    points stopped at 256 until round 4 (now 1024, like ILRMA)."""
    from oracle.gmnmf import GaussMNMFOracle
    from oracle.mnmf import FastGaussMNMFOracle
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    M, F, T = 3, 9, 24
    X = nmf_mixture(88, M, F, T)
    rng = np.random.default_rng(K)
    if cls == "fast":
        kw = dict(basis=rng.random((M, F, K)), activation=rng.random((M, K, T)),
                  spatial=rng.random((F, M, M)))
        ref = FastGaussMNMFOracle(n_basis=K)
        m = FastGaussMNMF(n_basis=K)
    elif cls == "gauss":
        kw = dict(basis=rng.random((M, F, K)), activation=rng.random((M, K, T)))
        ref = GaussMNMFOracle(n_basis=K, rng=np.random.default_rng(1))
        m = GaussMNMF(n_basis=K, rng=np.random.default_rng(1))
    else:
        latent = rng.random((M, K))
        kw = dict(basis=rng.random((F, K)), activation=rng.random((K, T)),
                  latent=latent / latent.sum(axis=0))
        ref = GaussMNMFOracle(n_basis=K, partitioning=True, rng=np.random.default_rng(1))
        m = GaussMNMF(n_basis=K, partitioning=True, rng=np.random.default_rng(1))
    Yr = ref.run(X, n_iter=3, **{k: v.copy() for k, v in kw.items()})
    Y = m(X, n_iter=3, **kw)
    np.testing.assert_allclose(m.loss, ref.loss, rtol=1e-8)
    assert rel_err(m.basis, ref.basis) < 1e-7 and rel_err(m.activation, ref.activation) < 1e-7
    assert rel_err(Y, Yr) < 1e-6


@pytest.mark.parametrize("M", [2, 3, 4, 6])
def test_to_psd_and_invsqrtmh_with_a_custom_floor_against_golden(M):
    """A flooring callable the kernels cannot run: eigen-decomposition on the device, the callable on
    the (..., M) eigenvalues on the host, P diag(.) P^H back on the device (ssspy_herm_rebuild).
    ref: ssspy/special/psd.py:11-71, ssspy/linalg/sqrtm.py:27-64."""
    from ssspy_amd.linalg import invsqrtmh
    from ssspy_amd.special import to_psd

    g = load_golden("psd_custom_floor")
    out = to_psd(g["m{}_H".format(M)], flooring_fn=_golden_custom_floor)
    assert rel_err(out, g["m{}_psd".format(M)]) < 1e-10
    assert np.all(np.linalg.eigvalsh(out) > 0)
    out = invsqrtmh(g["m{}_Hp".format(M)], flooring_fn=_golden_custom_floor)
    assert rel_err(out, g["m{}_invsqrt".format(M)]) < 1e-9


_PACKED_CASES = [(4, 4, "max"), (5, 3, "max"), (6, 6, "add"), (7, 7, "none"), (8, 8, "max"),
                 (8, 5, "tiny"), (5, 5, "tiny")]


@pytest.mark.parametrize("M,N,floor", _PACKED_CASES)
def test_gauss_mnmf_packed_route_against_oracle(M, N, floor):
    """The packed per-point kernels of GaussMNMF at 4-8 channels (in-place Cholesky inverse, one
    eigen-decomposition per spatial update, floors applied in the kernel, flag-gated repair by the
    full-storage kernels) against the oracle's restatement of ssspy/bss/mnmf.py:838-1073, a batch of
    three mixtures, four iterations.  "tiny": silent frames and a rank-deficient bin, where the
    eigenvalue floor acts and the repair kernels run.  (Rounds 4-5 compared the packed route with
    the full-storage one in a second process, SSSPY_AMD_GMNMF_FULL; the reference-generated
    gmnmf_floor_* goldens and this test replace that switch.)"""
    from oracle.gmnmf import GaussMNMFOracle
    from ssspy_amd.bss.mnmf import GaussMNMF
    from ssspy_amd.special.flooring import add_flooring, max_flooring

    B, F, T, K = 3, 33, 40, 3
    rng = np.random.default_rng(7)
    X = rng.standard_normal((B, M, F, T)) + 1j * rng.standard_normal((B, M, F, T))
    if floor == "tiny":
        X[:, :, :, : T // 4] *= 1e-9
        X[:, 1:, F // 2] = X[:, :1, F // 2]
    fn = {"max": functools.partial(max_flooring, eps=1e-10),
          "tiny": functools.partial(max_flooring, eps=1e-10),
          "add": functools.partial(add_flooring, eps=1e-8), "none": None}[floor]
    spec = {"max": ("max", 1e-10), "tiny": ("max", 1e-10), "add": ("add", 1e-8),
            "none": ("none", 0.0)}[floor]
    basis, act = rng.random((B, N, F, K)) + 0.05, rng.random((B, N, K, T)) + 0.05
    m = GaussMNMF(n_basis=K, n_sources=N, flooring_fn=fn)
    Y = m(X, n_iter=4, basis=basis.copy(), activation=act.copy())
    loss = np.asarray(m.loss)
    assert np.isfinite(Y).all() and np.isfinite(loss).all()
    for b in (0, 2):
        ref = GaussMNMFOracle(n_basis=K, n_sources=N, flooring=spec)
        Yr = ref.run(X[b], n_iter=4, basis=basis[b].copy(), activation=act[b].copy())
        np.testing.assert_allclose(loss[:, b], ref.loss, rtol=1e-7)
        for name in ("basis", "activation", "spatial"):
            assert rel_err(getattr(m, name)[b], getattr(ref, name)) < 1e-6, (b, name)
        assert rel_err(Y[b], Yr) < 1e-6, b


def test_page_locked_downloads_are_counted_and_capped(monkeypatch):
    """Round-5 advisor finding: results of 1-256 MB are handed out as page-locked blocks; a caller who
    keeps them must not pin memory without bound.  The blocks alive are counted at the size the
    host allocator rounds them to, capped (pageable copies past the cap) and released when the
    arrays die."""
    import gc

    import torch

    from ssspy_amd import _device as dv

    gc.collect()
    base = dv.pinned_outstanding_bytes()
    monkeypatch.setattr(dv, "_PINNED_OUTSTANDING_CAP", base + (8 << 20))
    t = torch.arange(3 << 17, dtype=torch.float64, device="cuda")  # 3 MiB -> a 4 MiB block
    kept = [dv.to_host(t) for _ in range(4)]
    assert dv.pinned_outstanding_bytes() - base == 8 << 20  # two blocks; the rest pageable
    assert [torch.from_numpy(a).is_pinned() for a in kept] == [True, True, False, False]
    for a in kept:
        np.testing.assert_array_equal(a, np.arange(3 << 17, dtype=np.float64))
    del kept, a
    gc.collect()
    assert dv.pinned_outstanding_bytes() == base


# ------------------------------------------------------------- the private copy of the mixture
@pytest.mark.parametrize("cls", ["ilrma", "auxiva", "fmnmf"])
def test_input_attribute_is_a_private_copy_formed_on_demand(cls):
    """The reference keeps ``self.input = input.copy()`` (ssspy/bss/ilrma.py:840, iva.py:152,
    mnmf.py:153).  Round 6: the private copy is the HBM buffer and ``input`` is formed from it on
    first access -- same values, same dtype, and untouched by what the caller does to the array
    afterwards; a call never forms it by itself."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(3, 3, 17, 40)
    for dtype in (np.complex128, np.complex64):
        Xin = X.astype(dtype)
        keep = Xin.copy()
        m = {"ilrma": lambda: GaussILRMA(n_basis=2, rng=np.random.default_rng(0)),
             "auxiva": lambda: AuxLaplaceIVA(),
             "fmnmf": lambda: FastGaussMNMF(n_basis=2, rng=np.random.default_rng(0))}[cls]()
        with pytest.raises(AssertionError, match="Specify data"):
            m._reset()
        Y = m(Xin, n_iter=2)
        assert m.__dict__.get("_input_value") is None  # (nothing on the path read it)
        Xin *= 0.0  # the caller reuses its buffer
        got = m.input
        assert got.dtype == dtype and got.shape == keep.shape
        assert np.array_equal(got, keep)
        assert m.input is got  # formed once
        assert Y.shape == keep.shape
    batch = np.stack([X, 2.0 * X])
    m = GaussILRMA(n_basis=2, rng=np.random.default_rng(0))
    m(batch, n_iter=1)
    assert np.array_equal(m.input, batch)
