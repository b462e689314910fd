"""Pins the CPU oracle (oracle/gml_oracle.c) against every golden vector the reference's own
tests hold for the learn() path (test/runtests.jl:66-102, 130-158; data in tests/golden/ are
verbatim copies of /root/reference/test/data/*.csv).

The goldens carry Ipopt's interior-point termination residual, so an exact optimiser lands
~1e-8 away on a/b/c and ~1e-4 away on mvt (lambda = 5.4e-5 there); SURVEY.md 8(c).  The
tolerances below are those measured gaps with head-room, plus a solver-independent KKT
certificate showing the oracle's answer is the true optimum.
"""
import numpy as np
import pytest

from conftest import DEFAULT_C, load_csv
from oracle import oracle as O

FORMS = ["RISE", "logRISE", "RPLE"]


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("form", FORMS)
def test_abc_goldens(name, form):
    # learn(samples, formulation) with default regulariser, symmetrised (runtests.jl:68-80)
    s = load_csv(f"{name}_samples.csv")
    R, kkt, _ = O.learn_pair(s, form, c=DEFAULT_C[form], symmetrize=True)
    G = load_csv(f"{name}_{form}_learned.csv")
    assert np.abs(R - G).max() <= 5e-8
    assert np.linalg.norm(R - G) / np.linalg.norm(G) <= 1e-6  # north-star tolerance
    assert kkt.max() <= 1e-10


@pytest.mark.parametrize("form", FORMS)
def test_mvt_goldens(form):
    # learn(samples, X(0.2, false), NLP(SOLVER))  (runtests.jl:83-101)
    s = load_csv("mvt_samples.csv")
    R, kkt, _ = O.learn_pair(s, form, c=0.2, symmetrize=False)
    G = load_csv(f"mvt_{form}_learned.csv")
    assert np.abs(R - G).max() <= 3e-4
    assert kkt.max() <= 1e-10  # certificate: this IS the optimum; the gap is Ipopt's barrier residual
    big = np.abs(G) > 0.01
    assert (np.abs(R - G)[big] / np.abs(G)[big]).max() <= 1e-3


@pytest.mark.parametrize("name", ["a", "b", "c", "mvt"])
def test_multirise_order2_equals_rise(name):
    # runtests.jl:132-158: multiRISE(0.2,false,2) == RISE(0.2,false) entry by entry, atol 1e-7
    s = load_csv(f"{name}_samples.csv")
    R, _, _ = O.learn_pair(s, "RISE", c=0.2, symmetrize=False)
    rec, kkt = O.learn_multi(s, c=0.2, symmetrize=False, order=2)
    n = R.shape[0]
    assert len(rec) == n * n
    for u in range(n):
        assert abs(rec[(u + 1,)] - R[u, u]) <= 1e-7
        for i in range(n):
            if i != u:
                assert abs(rec[(u + 1, i + 1)] - R[u, i]) <= 1e-7
    assert kkt.max() <= 1e-10


def test_lambda_formula():
    # :157  lambda = c*sqrt(log(n^2/0.05)/M); values quoted in SURVEY.md 8(a) I3
    assert O.lam(0.4, 1024, 1e6) == pytest.approx(1.64e-3, rel=5e-3)
    assert O.lam(0.2, 9, 1e8) == pytest.approx(5.44e-5, rel=5e-3)
    assert O.lam(0.4, 3, 1e6) == pytest.approx(9.12e-4, rel=5e-3)


@pytest.mark.parametrize("form", FORMS)
def test_objgrad_known_answers_at_zero(form):
    # theta = 0: RISE f = 1, g_i = -<s_u s~_i>; logRISE f = 0 same g; RPLE f = log 2, g = -<..>
    s = load_csv("mvt_samples.csv")
    counts, spins = O.split_histogram(s)
    w = counts / counts.sum()
    n = spins.shape[1]
    for u in (0, 4, 8):
        stat = spins[:, [u]] * spins.astype(float)
        stat[:, u] = spins[:, u]
        corr = (w[:, None] * stat).sum(0)
        f, g = O.objgrad_pair(s, form, u, np.zeros(n))
        f0 = {"RISE": 1.0, "logRISE": 0.0, "RPLE": np.log(2.0)}[form]
        assert f == pytest.approx(f0, abs=1e-13)
        np.testing.assert_allclose(g, -corr, atol=1e-13)


def test_objgrad_matches_numpy_restatement():
    # independent numpy statement of :191-208 at a pseudo-random theta, all three pointwise forms
    rng = np.random.default_rng(0)
    s = load_csv("mvt_samples.csv")
    counts, spins = O.split_histogram(s)
    w = counts / counts.sum()
    n = spins.shape[1]
    for u in range(n):
        th = rng.normal(scale=0.3, size=n)
        stat = spins[:, [u]] * spins.astype(float)
        stat[:, u] = spins[:, u]
        E = stat @ th
        e = w * np.exp(-E)
        ref = {
            "RISE": (e.sum(), -(stat * e[:, None]).sum(0)),
            "logRISE": (np.log(e.sum()), -(stat * e[:, None]).sum(0) / e.sum()),
            "RPLE": ((w * np.log1p(np.exp(-2 * E))).sum(), -(stat * (2 * w / (1 + np.exp(2 * E)))[:, None]).sum(0)),
        }
        for form, (f0, g0) in ref.items():
            f, g = O.objgrad_pair(s, form, u, th)
            assert f == pytest.approx(f0, rel=1e-13)
            np.testing.assert_allclose(g, g0, rtol=1e-11, atol=1e-14)


def test_fast_rise_nodes_matches_generic():
    rng = np.random.default_rng(1)
    s = load_csv("mvt_samples.csv")
    counts, spins = O.split_histogram(s)
    n = spins.shape[1]
    nodes = np.array([0, 3, 8])
    th = rng.normal(scale=0.2, size=(3, n))
    f, g = O.objgrad_rise_nodes(counts, spins, nodes, th)
    for a, u in enumerate(nodes):
        f0, g0 = O.objgrad_pair(s, "RISE", u, th[a])
        assert f[a] == pytest.approx(f0, rel=1e-13)
        np.testing.assert_allclose(g[a], g0, rtol=1e-11, atol=1e-14)


def test_multi_keys_order_and_count():
    # :94-104 + models.jl:228-246: (u), then (u,i) ascending, then (u,i,j) i<j lexicographic
    keys = O.multi_keys(4, 3, 1)
    assert keys == [(1,), (1, 0), (1, 2), (1, 3), (1, 0, 2), (1, 0, 3), (1, 2, 3)]
    assert len(O.multi_keys(9, 4, 0)) == 1 + 8 + 28 + 56


def test_multi_objgrad_matches_numpy():
    rng = np.random.default_rng(2)
    s = load_csv("c_samples.csv")
    counts, spins = O.split_histogram(s)
    w = counts / counts.sum()
    n = spins.shape[1]
    for u in range(n):
        keys = O.multi_keys(n, 3, u)
        th = rng.normal(scale=0.3, size=len(keys))
        stat = np.stack([np.prod(spins[:, list(k)].astype(float), axis=1) for k in keys], axis=1)
        e = w * np.exp(-(stat @ th))
        f, g = O.objgrad_multi(s, 3, u, th)
        assert f == pytest.approx(e.sum(), rel=1e-13)
        np.testing.assert_allclose(g, -(stat * e[:, None]).sum(0), rtol=1e-11, atol=1e-14)


# ---- gml_oracle_fast.c: the blocked / OpenMP restatements used for full-size checks and the CPU baseline ----

@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("form", FORMS)
def test_fast_learn_abc_goldens(name, form):
    # the batched working-set Newton (the device solver's algorithm on the CPU) against the reference's goldens
    s = load_csv(f"{name}_samples.csv")
    counts, spins = O.split_histogram(s)
    R, kkt, st = O.learn_pair_fast(counts, spins, form, c=DEFAULT_C[form], tol=1e-12)
    R = 0.5 * (R + R.T)  # :184-186
    G = load_csv(f"{name}_{form}_learned.csv")
    assert np.abs(R - G).max() <= 5e-8
    assert np.linalg.norm(R - G) / np.linalg.norm(G) <= 1e-6
    assert kkt.max() <= 1e-10


@pytest.mark.parametrize("form", FORMS)
def test_fast_learn_mvt_matches_dense_oracle(form):
    s = load_csv("mvt_samples.csv")
    counts, spins = O.split_histogram(s)
    R, kkt, _ = O.learn_pair_fast(counts, spins, form, c=0.2, tol=1e-12)
    R0, kkt0, _ = O.learn_pair(s, form, c=0.2, symmetrize=False)
    assert kkt.max() <= 1e-10
    assert np.abs(R - R0).max() <= 1e-8
    assert np.abs(R - load_csv(f"mvt_{form}_learned.csv")).max() <= 3e-4


@pytest.mark.parametrize("form", FORMS)
def test_fast_objgrad_nodes_matches_generic(form):
    # more than one 32-node block, ragged last block, repeated nodes, odd K, non-uniform counts
    rng = np.random.default_rng(4)
    K, n = 1001, 70
    spins = np.where(rng.random((K, n)) < 0.5, 1, -1).astype(np.int8)
    counts = rng.integers(0, 5, K).astype(float)
    hist = np.column_stack([counts, spins])
    nodes = np.concatenate([np.arange(n), [3, 3, 69]])
    th = rng.normal(scale=0.1, size=(len(nodes), n))
    f, g = O.objgrad_nodes(form, counts, spins, nodes, th)
    for a in (0, 31, 32, 33, 69, 70, 72):
        f0, g0 = O.objgrad_pair(hist, form, int(nodes[a]), th[a])
        assert f[a] == pytest.approx(f0, rel=1e-12, abs=1e-13)
        np.testing.assert_allclose(g[a], g0, rtol=1e-10, atol=1e-13)
    f1, _ = O.objgrad_nodes(form, None, spins, nodes[:2], th[:2], want_grad=False)  # counts = None: all ones
    f2, _ = O.objgrad_nodes(form, np.ones(K), spins, nodes[:2], th[:2])
    assert np.allclose(f1, f2, rtol=1e-14)


def test_fast_multi3_matches_generic():
    rng = np.random.default_rng(5)
    K, n = 403, 9
    spins = np.where(rng.random((K, n)) < 0.5, 1, -1).astype(np.int8)
    counts = rng.integers(1, 4, K).astype(float)
    hist = np.column_stack([counts, spins])
    P = 1 + 8 + 28
    nodes = np.array([0, 4, 8])
    th = rng.normal(scale=0.2, size=(3, P))
    f, g = O.objgrad_multi3_nodes(counts, spins, nodes, th)
    for a, u in enumerate(nodes):
        f0, g0 = O.objgrad_multi(hist, 3, int(u), th[a])
        assert f[a] == pytest.approx(f0, rel=1e-13)
        np.testing.assert_allclose(g[a], g0, rtol=1e-11, atol=1e-14)
