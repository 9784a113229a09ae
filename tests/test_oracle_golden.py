"""Pin the CPU oracle against the golden vectors generated from the reference.

Every fixture under tests/golden/ (made by tests/golden/make_golden.py, which
imports the reference in the build container) is replayed through ``oracle/``.
Tolerances: the oracle follows the reference's expression structure, so state
after 1/2/10 iterations must agree to ~1e-12 relative (Frobenius).
"""

import numpy as np
import pytest

from conftest import load_golden, rel_err, rel_err_up_to_phase
from conftest import option as _option
from oracle import spatial as sp
from oracle.ilrma import GaussILRMAOracle
from oracle.iva import AuxIVAOracle
from oracle.gmnmf import GaussMNMFOracle
from oracle.mnmf import FastGaussMNMFOracle

TOL = 1e-11

ILRMA_CASES = [
    "gilrma_ip1_n2", "gilrma_ip1_n3", "gilrma_ip1_n4", "gilrma_ip1_n4_p1", "gilrma_ip1_n2_add",
    "gilrma_ip1_n2_nofloor", "gilrma_ip1_n3_raw", "gilrma_iss1_n2", "gilrma_iss1_n4",
    "gilrma_iss1_n3_p1", "gilrma_ip2_n3", "gilrma_ip2_n2", "gilrma_iss2_n4", "gilrma_iss2_n3",
    "tilrma_ip1_n3", "tilrma_iss1_n2_p1", "tilrma_ip2_n3", "ggdilrma_ip1_n3", "ggdilrma_iss1_n2",
    "ggdilrma_iss2_n3_p1", "tilrma_iss2_n4", "ggdilrma_iss2_n4", "tilrma_iss2_n6_p1",
    "gilrma_me_ip1_n3", "tilrma_me_iss1_n2", "gilrma_part_ip1_n3", "gilrma_part_iss1_n2_p1",
    "gilrma_part_me_ip2_n3", "tilrma_part_ip1_n2", "ggdilrma_part_iss1_n3",
    "tilrma_part_me_nonorm_n2", "gilrma_ipa_n3", "gilrma_ipa_n2_p1", "gilrma_ipa_part_n4",
    "gilrma_ipa_newton8_n3",
    "gilrma_mdp_ip1_n3", "gilrma_mdp_iss1_n2", "gilrma_pbnorm_ip1_n3", "gilrma_pbnorm_iss1_n2_p1",
    "gilrma_ip1_n10", "gilrma_iss1_n9_p1",  # above 8 sources: the run-time-N kernels (wide_n.hip)
    "gilrma_ip2_n9", "gilrma_iss2_n10", "gilrma_ipa_n9", "gilrma_ipa_n12_add",
    # 3 / 4 sources, >= 16 frames per source (round 6: the implied-filter route of the device build)
    "gilrma_iss1_n4_t80", "gilrma_iss2_n4_t72", "gilrma_iss2_n3_t64", "gilrma_ipa_n3_t56",
    "gilrma_ipa_n4_t72",
]


def _model(g):
    kind = str(g["meta_model"]) if "meta_model" in g else "gauss"
    return (kind, None if kind == "gauss" else float(g["meta_model_param"]))
IVA_CASES = [
    "auxlap_ip1_n2", "auxlap_ip1_n4", "auxlap_iss1_n2", "auxlap_iss1_n8", "auxgauss_ip1_n3",
    "auxgauss_iss1_n3", "auxlap_ip1_n2_raw", "auxlap_ip2_n3", "auxlap_iss2_n4", "auxgauss_ip2_n2",
    "auxgauss_iss2_n3", "auxlap_ipa_n3", "auxgauss_ipa_n2", "auxlap_mdp_ip1_n3",
    "auxlap_mdp_iss1_n2",
    "auxlap_iss1_n12", "auxlap_ip1_n9", "auxgauss_ip1_n16_mdp",  # above 8 sources (wide_n.hip)
    "auxlap_ip2_n9", "auxlap_iss2_n12", "auxlap_ipa_n10",
    "auxlap_iss2_n4_t72", "auxlap_ipa_n3_t60", "auxgauss_ipa_n4_t70",  # (round 6, see above)
]
MNMF_CASES = ["fmnmf_ip1_m3", "fmnmf_ip1_m4", "fmnmf_ip1_m3_n2", "fmnmf_ip1_m2_nonorm",
              "fmnmf_ip1_m6_n3", "fmnmf_ip1_m5", "fmnmf_ip1_m8_n2"]


def _floor(g):
    return (str(g["meta_floor_kind"]), float(g["meta_floor_eps"]))


def _check_snapshots(g, k, model, names):
    pairwise = str(g["meta_algo"]) in ("IP2", "ISS2") if "meta_algo" in g else False
    for name in names:
        key = "it{}_{}".format(k, name)
        if key in g:
            if pairwise and name in ("demix_filter", "output"):
                assert rel_err_up_to_phase(getattr(model, name), g[key], name) < 1e-9, key
            else:
                # heavy-tailed models with phase-ambiguous pairwise updates amplify rounding more
                tol = 1e-9 if pairwise else TOL
                assert rel_err(getattr(model, name), g[key]) < tol, key


@pytest.mark.parametrize("case", ILRMA_CASES)
def test_gauss_ilrma(case):
    g = load_golden(case)
    m = GaussILRMAOracle(
        n_basis=int(g["meta_n_basis"]), spatial_algorithm=str(g["meta_algo"]),
        domain=float(g["meta_domain"]), flooring=_floor(g),
        normalization=_option(g["meta_normalization"]),
        scale_restoration=_option(g["meta_scale_restoration"]), model=_model(g),
        source_algorithm=str(g["meta_source_algorithm"]) if "meta_source_algorithm" in g else "MM",
        partitioning=bool(g["meta_partitioning"]) if "meta_partitioning" in g else False,
    )
    m.newton_iter = int(g["meta_newton_iter"])
    init = dict(basis=g["basis0"], activation=g["activation0"])
    if m.partitioning:
        init["latent"] = g["latent0"]
    m.reset(g["X"], **init)
    losses = [m.compute_loss()]
    for k in range(1, int(g["meta_n_iter"]) + 1):
        m.update_once()
        losses.append(m.compute_loss())
        _check_snapshots(g, k, m, ["demix_filter", "output", "basis", "activation", "latent"])
    np.testing.assert_allclose(losses, g["loss"], rtol=1e-10)
    if m.scale_restoration:
        m.restore_scale()
    # the pairwise updates take a different rounding path whenever an eigenvector phase differs
    final_tol = 1e-8 if str(g["meta_algo"]) in ("IP2", "ISS2") else TOL
    if m.demix_filter is not None:
        m.output = sp.separate(m.input, m.demix_filter)
        assert rel_err(m.demix_filter, g["final_demix_filter"]) < final_tol
    assert rel_err(m.output, g["final_output"]) < final_tol


def test_gauss_ilrma_kat1_scalars():
    """KAT-1 of SURVEY.md section 8c, via run()."""
    g = load_golden("gilrma_ip1_n2")
    m = GaussILRMAOracle(n_basis=2, spatial_algorithm="IP")
    Y = m.run(g["X"], n_iter=10, basis=g["basis0"], activation=g["activation0"])
    assert m.loss[0] == pytest.approx(251.143395021126, rel=1e-11)
    assert m.loss[10] == pytest.approx(53.191617258319, rel=1e-11)
    assert Y[0, 0, 0] == pytest.approx(0.19347346183741757 - 0.2616337924299037j, rel=1e-10)
    assert np.sum(np.abs(Y) ** 2) == pytest.approx(1107.737755038587, rel=1e-11)


@pytest.mark.parametrize("case", IVA_CASES)
def test_aux_iva(case):
    g = load_golden(case)
    m = AuxIVAOracle(
        spatial_algorithm=str(g["meta_algo"]), contrast=str(g["meta_contrast"]),
        flooring=_floor(g), scale_restoration=_option(g["meta_scale_restoration"]),
    )
    m.reset(g["X"])
    losses = [m.compute_loss()]
    for k in range(1, int(g["meta_n_iter"]) + 1):
        m.update_once()
        losses.append(m.compute_loss())
        _check_snapshots(g, k, m, ["demix_filter", "output", "variance"])
    np.testing.assert_allclose(losses, g["loss"], rtol=1e-10)
    if m.scale_restoration:
        m.restore_scale()
    if m.demix_filter is not None:
        m.output = sp.separate(m.input, m.demix_filter)
        assert rel_err(m.demix_filter, g["final_demix_filter"]) < TOL
    assert rel_err(m.output, g["final_output"]) < TOL


@pytest.mark.parametrize("case", ["auxgeneric_ip1_n3", "auxgeneric_iss1_n2", "auxgeneric_ip2_n3"])
def test_generic_aux_iva_with_user_closures(case):
    """The reference's generic AuxIVA class driven by user closures (G_R(r) = r^1.5)."""
    g = load_golden(case)
    algo = str(g["meta_algo"])
    m = AuxIVAOracle(spatial_algorithm=algo, contrast=("power", float(g["meta_power"])))
    m.reset(g["X"])
    losses = [m.compute_loss()]
    for k in range(1, int(g["meta_n_iter"]) + 1):
        m.update_once()
        losses.append(m.compute_loss())
        if algo != "IP2":
            _check_snapshots(g, k, m, ["demix_filter", "output"])
    np.testing.assert_allclose(losses, g["loss"], rtol=1e-10)
    m.restore_scale()
    if m.demix_filter is not None:
        m.output = sp.separate(m.input, m.demix_filter)
    assert rel_err(m.output, g["final_output"]) < 1e-9


def test_all_channel_scale_restoration():
    """projection_back / minimal_distortion_principle with reference_id=None."""
    g = load_golden("restoration_all_channels")
    for N in (2, 3, 4):
        X, Y, W = (g["n{}_{}".format(N, k)] for k in "XYW")
        assert rel_err(sp.projection_back_filter(W, None), g["n{}_pb_filter".format(N)]) < TOL
        assert rel_err(sp.projection_back_output(Y, X, None), g["n{}_pb_output".format(N)]) < TOL
        assert rel_err(sp.minimal_distortion_output(Y, X, None), g["n{}_mdp_output".format(N)]) < TOL


@pytest.mark.parametrize("case,algo", [("kat_auxlap_ip1_config1", "IP"), ("kat_auxlap_iss1_config1", "ISS")])
def test_aux_iva_config1_kat(case, algo):
    """BASELINE.json configs[0] (N=2, F=257, T=128, 10 it): known-answer scalars."""
    g = load_golden(case)
    N, F, T = (int(v) for v in g["meta_shape"])
    rng = np.random.default_rng(int(g["meta_seed"]))
    X = rng.standard_normal((N, F, T)) + 1j * rng.standard_normal((N, F, T))
    m = AuxIVAOracle(spatial_algorithm=algo, contrast="laplace")
    Y = m.run(X, n_iter=int(g["meta_n_iter"]))
    np.testing.assert_allclose(m.loss, g["loss"], rtol=1e-10)
    assert Y[0, 0, 0] == pytest.approx(complex(g["kat_y000"]), rel=1e-9)
    assert np.sum(np.abs(Y) ** 2) == pytest.approx(float(g["kat_energy"]), rel=1e-10)


MNMF_IP2_CASES = ["fmnmf_ip2_m2", "fmnmf_ip2_m3", "fmnmf_ip2_m4", "fmnmf_ip2_m3_n2",
                  "fmnmf_ip2_m4_comb", "fmnmf_ip2_m5"]


def mnmf_pairs(g):
    """The pair list of a FastGaussMNMF-IP2 fixture (None: the reference's sequential default)."""
    if "meta_pairs" in g and str(g["meta_pairs"]) == "combination":
        return sp.combination_pairs(g["X"].shape[0])
    return None


@pytest.mark.parametrize("case", MNMF_CASES + MNMF_IP2_CASES)
def test_fast_gauss_mnmf(case):
    """IP1 and IP2 (ssspy/bss/mnmf.py:1449-1633) diagonaliser updates.  With IP2 the rows of Q carry
    the arbitrary phase of the 2x2 generalised eigenvectors, so Q is compared up to one phase per
    (bin, row); everything else -- D, T, V, the loss, the Wiener-filter output -- is phase-free."""
    g = load_golden(case)
    n_sources = int(g["meta_n_sources"])
    ip2 = "meta_diag_algo" in g and str(g["meta_diag_algo"]) == "IP2"
    m = FastGaussMNMFOracle(
        n_basis=int(g["meta_n_basis"]), n_sources=n_sources, flooring=_floor(g),
        normalization=_option(g["meta_normalization"]),
        diagonalizer_algorithm="IP2" if ip2 else "IP", pairs=mnmf_pairs(g),
    )
    m.reset(g["X"], basis=g["basis0"], activation=g["activation0"], spatial=g["spatial0"].copy())
    losses = [m.compute_loss()]
    for k in range(1, int(g["meta_n_iter"]) + 1):
        m.update_once()
        losses.append(m.compute_loss())
        if ip2:
            key = "it{}_diagonalizer".format(k)
            if key in g:
                assert rel_err_up_to_phase(m.diagonalizer, g[key], "demix_filter") < 1e-9, key
            _check_snapshots(g, k, m, ["spatial", "basis", "activation"])
        else:
            _check_snapshots(g, k, m, ["diagonalizer", "spatial", "basis", "activation"])
    np.testing.assert_allclose(losses, g["loss"], rtol=1e-10)
    Y = m.separate(m.input)
    assert rel_err(Y, g["final_output"]) < 1e-9


@pytest.mark.parametrize("N", [2, 3, 4, 5, 8])
def test_pairwise_operators(N):
    """update_by_ip2 / update_by_iss2 against the reference's outputs
    (ssspy/bss/_update_spatial_model.py:81-143, 197-314), up to the eigenvector phase per (bin, row)
    resp. (bin, source)."""
    g = load_golden("pairwise_operators")
    p = lambda s: g["n{}_".format(N) + s]  # noqa: E731
    W, U, Y, varphi = p("W"), p("U"), p("Y"), p("varphi")
    tol = 1e-10
    assert rel_err_up_to_phase(sp.update_by_ip2(W, U), p("ip2_out"), "demix_filter") < tol
    assert rel_err_up_to_phase(sp.update_by_ip2(W, U, pairs=sp.combination_pairs(N)),
                               p("ip2_out_comb"), "demix_filter") < tol
    assert rel_err_up_to_phase(sp.update_by_ip2(W, U, ("add", 1e-3)), p("ip2_out_add"),
                               "demix_filter") < tol
    pairs = [tuple(int(v) for v in pr) for pr in p("ip2_pairs")]
    assert rel_err_up_to_phase(sp.update_by_ip2(W, U, pairs=pairs), p("ip2_out_pairs"),
                               "demix_filter") < tol
    assert np.array_equal(p("ip2_out_copy"), p("ip2_out"))
    assert rel_err_up_to_phase(sp.update_by_iss2(Y, varphi), p("iss2_out"), "output") < tol
    assert rel_err_up_to_phase(sp.update_by_iss2(Y, varphi, pairs=sp.combination_pairs(N)),
                               p("iss2_out_comb"), "output") < tol
    assert rel_err_up_to_phase(sp.update_by_iss2(Y, varphi[:, :1, :], ("add", 1e-3)),
                               p("iss2_out_bcast_add"), "output") < tol
    pairs = [tuple(int(v) for v in pr) for pr in p("iss2_pairs")]
    assert rel_err_up_to_phase(sp.update_by_iss2(Y, varphi, pairs=pairs), p("iss2_out_pairs"),
                               "output") < tol


GMNMF_CASES = ["gmnmf_m2", "gmnmf_m3", "gmnmf_m4_n3", "gmnmf_m2_nonorm_add", "gmnmf_part_m3",
               "gmnmf_part_m2_n3", "gmnmf_m5", "gmnmf_m6_n3", "gmnmf_m7", "gmnmf_m8",
               # eigenvalue floor of to_psd active at most points (eps = 0.3), 10 iterations
               "gmnmf_floor_m5", "gmnmf_floor_m6_n3", "gmnmf_floor_m8"]


@pytest.mark.parametrize("case", GMNMF_CASES)
def test_gauss_mnmf(case):
    """Full-rank MNMF: Hermitian eigen-floors and a matrix geometric mean sit in the loop, so the
    restatement is held to 1e-8 (the reference's own regression tolerance is atol 1e-7)."""
    g = load_golden(case)
    m = GaussMNMFOracle(
        n_basis=int(g["meta_n_basis"]), n_sources=int(g["meta_n_sources"]), flooring=_floor(g),
        normalization=_option(g["meta_normalization"]),
        partitioning=bool(g["meta_partitioning"]) if "meta_partitioning" in g else False,
    )
    init = dict(basis=g["basis0"], activation=g["activation0"])
    if "spatial0" in g:
        init["spatial"] = g["spatial0"]
    if m.partitioning:
        init["latent"] = g["latent0"]
    m.reset(g["X"], **init)
    losses = [m.compute_loss()]
    for k in range(1, int(g["meta_n_iter"]) + 1):
        m.update_once()
        losses.append(m.compute_loss())
        for name in ("spatial", "basis", "activation", "latent"):
            key = "it{}_{}".format(k, name)
            if key in g:
                assert rel_err(getattr(m, name), g[key]) < 1e-8, key
    np.testing.assert_allclose(losses, g["loss"], rtol=1e-9)
    assert rel_err(m.separate(m.input), g["final_output"]) < 1e-8


@pytest.mark.parametrize("N", [2, 3, 4, 8])
def test_operators(N):
    g = load_golden("operators")
    p = lambda s: g[s.format(N)]  # noqa: E731
    assert rel_err(sp.update_by_ip1(p("ip1_n{}_W"), p("ip1_n{}_U")), p("ip1_n{}_out")) < TOL
    assert rel_err(sp.update_by_ip1(p("ip1_n{}_W"), p("ip1_n{}_U"), ("add", 1e-3)), p("ip1_n{}_out_add")) < TOL
    assert rel_err(sp.update_by_iss1(p("iss1_n{}_Y"), p("iss1_n{}_varphi")), p("iss1_n{}_out")) < TOL
    assert rel_err(sp.update_by_iss1(p("iss1_n{}_Y"), p("iss1_n{}_varphi")[:, :1, :]), p("iss1_n{}_out_bcast")) < TOL
    assert rel_err(sp.projection_back_filter(p("ip1_n{}_W"), 1), p("pb_n{}_filter")) < TOL
    assert rel_err(sp.projection_back_output(p("iss1_n{}_Y"), p("pb_n{}_X"), 0), p("pb_n{}_output")) < TOL
    assert rel_err(sp.to_psd(p("psd_n{}_in")), p("psd_n{}_out")) < TOL


@pytest.mark.parametrize("N", [2, 3, 4, 5])
def test_ipa_operator(N):
    from oracle.ipa import update_by_ipa

    g = load_golden("ipa_operators")
    Y, varphi = g["n{}_Y".format(N)], g["n{}_varphi".format(N)]
    assert rel_err(update_by_ipa(Y, varphi), g["n{}_out".format(N)]) < 1e-11
    assert rel_err(update_by_ipa(Y, varphi, normalization=False, max_iter=3),
                   g["n{}_out_nonorm_it3".format(N)]) < 1e-11
    assert rel_err(update_by_ipa(Y, varphi[:, :1, :], flooring=("add", 1e-4)),
                   g["n{}_out_bcast_add".format(N)]) < 1e-11
    # twelve steps allowed: the reference's loop stops as soon as every bin has converged
    assert rel_err(update_by_ipa(Y, varphi, max_iter=12), g["n{}_out_it12".format(N)]) < 1e-11


@pytest.mark.parametrize("N", [7, 8])
def test_ipa_operator_seven_and_eight_sources(N):
    """Round 5: update_by_ipa of the reference at the source counts the device build runs with a bin
    on 8 lanes (fixture eight_lane_operators, generated by tests/golden/make_golden.py)."""
    import warnings

    from oracle.ipa import update_by_ipa

    g = load_golden("eight_lane_operators")
    Y, varphi = g["ipa{}_Y".format(N)], g["ipa{}_varphi".format(N)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert rel_err(update_by_ipa(Y, varphi), g["ipa{}_out".format(N)]) < 1e-10
        assert rel_err(update_by_ipa(Y, varphi, normalization=False, max_iter=3),
                       g["ipa{}_out_nonorm_it3".format(N)]) < 1e-10
        assert rel_err(update_by_ipa(Y, varphi[:, :1, :], flooring=("add", 1e-4)),
                       g["ipa{}_out_bcast_add".format(N)]) < 1e-10
        assert rel_err(update_by_ipa(Y, varphi, max_iter=12), g["ipa{}_out_it12".format(N)]) < 1e-10


@pytest.mark.parametrize("L", [1, 2, 3, 5])
def test_lqpqm2_operator(L):
    """ssspy.linalg.lqpqm2 on a batch: the Newton loop stops when ALL problems have converged
    (lqpqm.py:196-213) and warns when they have not."""
    import warnings

    from oracle.ipa import lqpqm2

    g = load_golden("ipa_operators")
    H, v, z = (g["lq{}_{}".format(L, k)] for k in "Hvz")
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # ten steps: converged, no warning
        assert rel_err(lqpqm2(H, v, z, ("max", 1e-10), 10), g["lq{}_y".format(L)]) < 1e-10
    assert rel_err(lqpqm2(H, v, z, ("max", 1e-10), 2), g["lq{}_y_it2".format(L)]) < 1e-10


def lqpqm_singular_check(y, ref, singular, tol=1e-10):
    """Non-singular problems: equal.  Singular ones: the reference returns scale * (last row of the
    eigenvector matrix), each entry carrying the arbitrary phase of a different eigenvector
    (lqpqm.py:84-93) -- only the moduli are defined."""
    assert rel_err(y[~singular], ref[~singular]) < tol
    if singular.any():
        assert rel_err(np.abs(y[singular]), np.abs(ref[singular])) < tol


@pytest.mark.parametrize("L", [1, 2, 3, 5])
def test_lqpqm2_singular_fn(L):
    """singular_fn = default / None / a callable (ssspy/linalg/lqpqm.py:61-78) on problems with v
    exactly zero, below the floor, and small."""
    from oracle import ipa

    g = load_golden("lqpqm_singular")
    H, v, z = (g["l{}_{}".format(L, k)] for k in ("H", "v", "z"))
    norms = np.linalg.norm(v, axis=-1)
    for key, kw, singular in (("y_default", {}, norms < 1e-10),
                              ("y_none", dict(singular_fn=None), norms == 0),
                              ("y_callable", dict(singular_fn=lambda x: x < 0.5), norms < 0.5)):
        ref = g["l{}_{}".format(L, key)]
        y = ipa.lqpqm2(H, v, z, ("max", 1e-10), 10, **kw)
        lqpqm_singular_check(y, ref, singular)


def test_inv2():
    g = load_golden("operators")
    assert rel_err(sp.inv2(g["inv2_in"]), g["inv2_out"]) < 1e-13


# --------------------------------------------------------------------------- rng-drawn initial state
def _golden_custom_floor(x):
    """Same function as tests/golden/make_golden.py:custom_floor (a fixture cannot carry code)."""
    return np.maximum(x, 1e-8) + 1e-12


RNG_INIT_CASES = ["rnginit_gilrma_n3", "rnginit_gilrma_part_n3", "rnginit_gilrma_part_iss_n2",
                  "rnginit_tilrma_part_n2", "rnginit_fmnmf_m3", "rnginit_gmnmf_m2",
                  "rnginit_gmnmf_part_m2"]
CUSTOM_FLOOR_CASES = ["customfloor_gilrma_ip1_n3", "customfloor_gilrma_iss1_n2",
                      "customfloor_auxlap_ip1_n3", "customfloor_auxlap_iss1_n2",
                      "customfloor_fmnmf_m3",
                      "customfloor_gilrma_ip2_n3", "customfloor_gilrma_iss2_n4",
                      "customfloor_gilrma_part_ip1_n3", "customfloor_gilrma_part_iss1_n2",
                      "customfloor_auxlap_ip2_n3", "customfloor_auxlap_iss2_n3",
                      "customfloor_auxgauss_ip1_n3", "customfloor_auxgauss_iss1_n2",
                      "customfloor_auxgauss_ip2_n3",
                      "customfloor_tilrma_ip2_n3", "customfloor_tilrma_iss2_n3",
                      "customfloor_ggdilrma_ip1_n3", "customfloor_ggdilrma_iss1_n2",
                      "customfloor_ggdilrma_ip2_n3", "customfloor_ggdilrma_iss2_n3"]


def _oracle_for(g, flooring=sp.DEFAULT_FLOOR):
    """Oracle instance for a ``rnginit_*`` / ``customfloor_*`` fixture: constructor arguments from
    the meta keys, the generator seeded like the reference run, NO injected state."""
    kind = str(g["meta_kind"]).split("_")[-1]
    rng = np.random.default_rng(int(g["meta_seed"]) + 3)
    K = int(g["meta_n_basis"])
    part = bool(g["meta_partitioning"]) if "meta_partitioning" in g else False
    if kind == "ilrma":
        model = (str(g["meta_model"]), float(g["meta_model_param"])) if "meta_model" in g else ("gauss", None)
        if model[0] == "gauss":
            model = ("gauss", None)
        return GaussILRMAOracle(n_basis=K, spatial_algorithm=str(g["meta_spatial_algorithm"]),
                                partitioning=part, model=model, rng=rng, flooring=flooring), \
            ["latent", "basis", "activation", "demix_filter", "output"]
    if kind in ("iva", "gaussiva"):
        return AuxIVAOracle(spatial_algorithm=str(g["meta_spatial_algorithm"]),
                            contrast="laplace" if kind == "iva" else "gauss",
                            flooring=flooring), ["demix_filter", "output", "variance"]
    if kind == "fmnmf":
        return FastGaussMNMFOracle(n_basis=K, rng=rng, flooring=flooring), \
            ["basis", "activation", "diagonalizer", "spatial"]
    return GaussMNMFOracle(n_basis=K, partitioning=part, rng=rng, flooring=flooring), \
        ["basis", "activation", "latent", "spatial"]


def _replay_uninjected(g, m, names, tol):
    m.reset(g["X"])
    n_iter = int(g["meta_n_iter"])
    losses = [m.compute_loss()]
    for k in range(n_iter + 1):
        if k > 0:
            m.update_once()
            losses.append(m.compute_loss())
        for name in names:
            key = "it{}_{}".format(k, name)
            if key in g and getattr(m, name, None) is not None:
                pairwise = "meta_spatial_algorithm" in g and \
                    str(g["meta_spatial_algorithm"]) in ("IP2", "ISS2")
                if k == 0 and name not in ("output", "variance"):
                    # the drawn state itself: the same generator calls in the same order
                    np.testing.assert_array_equal(getattr(m, name), g[key], err_msg=key)
                elif pairwise and k > 0 and name in ("demix_filter", "output"):
                    # eigenvector phase of the pairwise updates, removed only by projection back
                    assert rel_err_up_to_phase(getattr(m, name), g[key], name) < max(tol, 1e-9), key
                else:
                    assert rel_err(getattr(m, name), g[key]) < (max(tol, 1e-9) if pairwise else tol), key
    np.testing.assert_allclose(losses, g["loss"], rtol=max(1e-10, tol))


@pytest.mark.parametrize("case", RNG_INIT_CASES)
def test_rng_drawn_initial_state(case):
    """Nothing injected: the parameters are drawn from ``rng`` in the reference's order
    (ssspy/bss/ilrma.py:230-266 latent -> basis -> activation under partitioning;
    mnmf.py:221-254 basis -> activation -> latent; mnmf.py:535-538 basis -> activation -> spatial)."""
    g = load_golden(case)
    m, names = _oracle_for(g)
    _replay_uninjected(g, m, names, 1e-8 if "gmnmf" in case else TOL)


@pytest.mark.parametrize("case", CUSTOM_FLOOR_CASES)
def test_custom_flooring_callable(case):
    """``flooring_fn`` may be any callable in the reference (ssspy/bss/ilrma.py:70-89)."""
    g = load_golden(case)
    m, names = _oracle_for(g, flooring=_golden_custom_floor)
    _replay_uninjected(g, m, names, TOL)


def test_fixture_key_sets():
    """Every fixture carries the keys the current generator writes for its kind (no stale files)."""
    import glob
    import os

    from conftest import GOLDEN_DIR

    need = {"gauss_ilrma": {"meta_source_algorithm", "meta_partitioning", "meta_model", "loss", "X"},
            "aux_iva": {"meta_algo", "meta_contrast", "loss"},
            "fast_gauss_mnmf": {"meta_n_sources", "loss", "spatial0"},
            "gauss_mnmf": {"meta_partitioning", "loss"}}
    seen = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        g = load_golden(os.path.basename(path)[:-4])
        kind = str(g["meta_kind"]) if "meta_kind" in g else None
        if kind in need:
            seen += 1
            assert need[kind] <= set(g), (path, need[kind] - set(g))
    assert seen >= 60


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/ssspy"),
                    reason="needs the reference checkout (build container only)")
def test_generator_reproduces_committed_fixtures(tmp_path):
    """tests/golden/make_golden.py writes deterministic archives: regenerating a sample of fixtures
    from the reference gives the committed bytes."""
    import hashlib
    import os
    import subprocess
    import sys

    from conftest import GOLDEN_DIR

    sample = ["gilrma_ip1_n4", "rnginit_gilrma_part_n3", "fmnmf_ip1_m4", "auxlap_iss1_n8", "operators"]
    env = dict(os.environ, SSSPY_GOLDEN_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(GOLDEN_DIR, "make_golden.py")] + sample, env=env,
                   check=True, capture_output=True, timeout=600)
    for name in sample:
        a = hashlib.sha256(open(os.path.join(GOLDEN_DIR, name + ".npz"), "rb").read()).hexdigest()
        b = hashlib.sha256(open(os.path.join(str(tmp_path), name + ".npz"), "rb").read()).hexdigest()
        assert a == b, name
