"""On-device exact sampler (the step before the learn() path) against the exact state
probabilities restated from src/sampling.jl:26-30, and the reference's sampler/accuracy tests
(test/runtests.jl:55-63, 105-127, 188-196) re-run with it."""
import numpy as np
import pytest

import gml_amd as gml
from conftest import MODELS

pytestmark = pytest.mark.gpu


def exact_probabilities(m):
    n = m.shape[0]
    states = ((np.arange(2 ** n)[:, None] >> np.arange(n)) & 1) * 2 - 1  # int_to_spin (sampling.jl:11-14)
    sf = states.astype(float)
    A = m - np.diag(np.diag(m))
    en = 0.5 * ((sf @ A) * sf).sum(1) + sf @ np.diag(m)  # weigh_proba (sampling.jl:26-30)
    p = np.exp(en - en.max())
    return states, p / p.sum()


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_sampler_matches_exact_distribution(name):
    m = MODELS[name]
    N = 1000000
    hist = gml.sample(gml.FactorGraph(m), N, seed=0)
    assert hist[:, 0].sum() == N  # runtests.jl:60
    states, p = exact_probabilities(m)
    lookup = {tuple(s): pi for s, pi in zip(states, p)}
    assert len(hist) == len(states)  # every state of these small models is observed at N = 1e6
    for row in hist:
        expect = lookup[tuple(row[1:])] * N
        assert abs(row[0] - expect) <= 6 * np.sqrt(expect) + 1  # 6 sigma of the binomial count


def test_sampler_replicates_and_seeds():
    m = MODELS["c"]
    reps = gml.sample(gml.FactorGraph(m), 20000, 3, seed=1)
    assert len(reps) == 3 and all(r[:, 0].sum() == 20000 for r in reps)  # runtests.jl:55-63
    assert not np.array_equal(reps[0], reps[1])
    again = gml.sample(gml.FactorGraph(m), 20000, seed=1)
    assert np.array_equal(again, reps[0])  # counter-based RNG: reproducible


def test_block_structured_model_beyond_enumeration():
    # 64 spins in 4 independent blocks of 16: 2^64 states overall, sampled exactly block by block
    synthetic = __import__("importlib").import_module("gml_amd.synthetic")
    _, J = synthetic.block_ising(64, 10, block=16, seed=5)
    with gml.Problem(model=J, num_samples=200000, seed=3) as p:
        assert (p.K, p.n, p.M) == (200000, 64, 200000.0)
        spins = p.spins()
        out, kkt, st = p.learn("RISE", 0.4, tol=1e-9)
    assert set(np.unique(spins)) == {-1, 1}
    # first block marginal distribution against the exact one
    states, pr = exact_probabilities(J[:16, :16])
    idx = ((spins[:, :16] > 0) * (1 << np.arange(16))).sum(1)
    emp = np.bincount(idx, minlength=2 ** 16) / len(idx)
    assert np.abs(emp - pr).max() <= 6 * np.sqrt(pr.max() / len(idx))
    assert np.abs(0.5 * (out + out.T) - J).max() <= 0.1  # and the learner recovers the model from them
    with pytest.raises(gml.GMLError):  # a 24-spin component cannot be enumerated
        big = np.zeros((24, 24))
        big[np.arange(23), np.arange(1, 24)] = 0.1
        gml.Problem(model=big + big.T, num_samples=10)


def test_learned_model_accuracy_with_device_sampler():
    # runtests.jl:105-127 and the docs example :188-196, samples drawn on the device
    for name, m in MODELS.items():
        for N, thr in ((1000, 0.15), (10000, 0.05)):
            hist = gml.sample(gml.FactorGraph(m), N, seed=0)
            for F in (gml.RISE, gml.logRISE, gml.RPLE):
                R = gml.learn(hist, F(), gml.HIP(tol=1e-9))
                assert np.abs(R - m).max() <= thr
    model = np.array([[0.0, 0.1, 0.2], [0.1, 0.0, 0.3], [0.2, 0.3, 0.0]])
    learned = gml.learn(gml.sample(gml.FactorGraph(model), 100000, seed=0))
    assert np.abs(learned - model).max() <= 0.01


def test_multibody_sampler_matches_exact_distribution():
    # the reference's general sampler (sampling.jl:60-88): P(s) ~ exp(sum_t w_t prod_{i in t} s_i) with 3- and
    # 4-body terms; two independent components, one of them beyond what the test can enumerate jointly
    terms = {(1,): 0.2, (2, 3): -0.4, (1, 2, 3): 0.5, (2, 3, 4): -0.3, (1, 2, 3, 4): 0.25, (4,): -0.1,
             (5, 6, 7): 0.6, (5,): 0.1, (6, 7): 0.3}
    fg = gml.FactorGraph(4, 7, "spin", terms)
    N = 1000000
    hist = gml.sample(fg, N, seed=2)
    assert hist[:, 0].sum() == N
    n = 7
    states = ((np.arange(2 ** n)[:, None] >> np.arange(n)) & 1) * 2 - 1
    en = np.zeros(2 ** n)
    for k, w in terms.items():
        en += w * np.prod(states[:, [i - 1 for i in k]], axis=1)
    pr = np.exp(en - en.max())
    pr /= pr.sum()
    lookup = {tuple(s): pi for s, pi in zip(states, pr)}
    assert len(hist) == 2 ** n
    for row in hist:
        expect = lookup[tuple(row[1:])] * N
        assert abs(row[0] - expect) <= 6 * np.sqrt(expect) + 1
    # and multiRISE recovers the terms from device-sampled data without a host round trip
    with gml.Problem(terms=terms, n=7, num_samples=400000, seed=5, order=4) as p:
        out, kkt, st = p.learn("RISE", 0.2, tol=1e-9)
        keys0 = p.multi_keys(0)
    got = {tuple(sorted(i + 1 for i in k)): v for k, v in zip(keys0, out[0])}
    for k in [(1,), (1, 2, 3), (1, 2, 3, 4)]:
        assert abs(got[k] - terms[k]) <= 0.03


def test_glauber_chains_match_exact_distribution_and_feed_the_learner():
    # (beyond the reference) heat-bath chains on the device: on a 3 x 4 open lattice with fields the empirical
    # distribution of 4e5 independent chains after 60 sweeps must match the exact one ...
    rng = np.random.default_rng(3)
    L1, L2 = 3, 4
    n = L1 * L2
    m = np.zeros((n, n))
    for a in range(L1):
        for b in range(L2):
            i = a * L2 + b
            if b + 1 < L2:
                m[i, i + 1] = m[i + 1, i] = rng.uniform(0.2, 0.5) * rng.choice([-1, 1])
            if a + 1 < L1:
                m[i, i + L2] = m[i + L2, i] = rng.uniform(0.2, 0.5) * rng.choice([-1, 1])
            m[i, i] = rng.uniform(-0.2, 0.2)
    N = 400000
    hist = gml.sample(gml.FactorGraph(m), N, sampler=gml.Glauber(60), seed=4)
    assert hist[:, 0].sum() == N
    states, p = exact_probabilities(m)
    lookup = {tuple(s): pi for s, pi in zip(states, p)}
    emp = np.zeros(len(states))
    idx = {tuple(s): i for i, s in enumerate(states)}
    for row in hist:
        emp[idx[tuple(row[1:])]] = row[0] / N
    assert np.abs(emp - p).max() <= 6 * np.sqrt(p.max() / N)
    # ... and a 16 x 16 periodic lattice (256 spins in ONE component: far beyond enumeration) sampled this way
    # lets RISE recover its couplings
    Lx = 16
    n = Lx * Lx
    J = np.zeros((n, n))
    for a in range(Lx):
        for b in range(Lx):
            i = a * Lx + b
            for j in (a * Lx + (b + 1) % Lx, ((a + 1) % Lx) * Lx + b):
                J[i, j] = J[j, i] = 0.3 * (1 if (a + b) % 3 else -1)
    terms = {(i + 1, j + 1): J[i, j] for i in range(n) for j in range(i + 1, n) if J[i, j] != 0}
    with gml.Problem(terms=terms, n=n, num_samples=300000, seed=9, mcmc_sweeps=150) as p:
        out, kkt, st = p.learn("RISE", 0.4, tol=1e-9, precision="i8x")
    assert st["not_converged"] == 0
    assert np.abs(0.5 * (out + out.T) - J).max() <= 0.05


def test_multibody_learn_sample_relearn_round_trip():
    # runtests.jl:161-181: learn an order-4 multiRISE model (lambda = 0, un-symmetrised) from 10000 samples, sample from the
    # LEARNED FactorGraph, learn again.  The un-symmetrised result holds every coupling once per member spin -- keys (1,2) and
    # (2,1) -- and the general sampler sums all terms (sampling.jl:60-65), so the second model's couplings come back twice as
    # large; the reference's test documents exactly that ("this is a bug ...") and checks value / 2 -- so does this one.
    from conftest import MODELS
    for name, mtx in MODELS.items():
        gm_tmp = gml.FactorGraph(mtx)
        order = min(4, mtx.shape[0])
        hist = gml.sample(gm_tmp, 10000, seed=0)
        learned_gm = gml.learn(hist, gml.multiRISE(0.0, False, order), gml.HIP(tol=1e-10))
        for key, value in gm_tmp:
            assert learned_gm[key] == pytest.approx(value, abs=0.15)
        assert learned_gm.order == order and len(learned_gm) > len(gm_tmp)
        hist2 = gml.sample(learned_gm, 10000, seed=1)
        learned_gm2 = gml.learn(hist2, gml.multiRISE(0.0, False, order), gml.HIP(tol=1e-10))
        for key, value in gm_tmp:
            assert learned_gm2[key] / 2.0 == pytest.approx(value, abs=0.16)


def _mvt_model(golden):
    m = golden("mvt_RISE_learned.csv")
    return 0.5 * (m + m.T)


def test_histogramming_on_the_device_1e8_draws_of_a_9_spin_model(golden):
    # sample(gm, N) = countmap of the draws (sampling.jl:52-54): sorted and run-length encoded on the device, so 1e8 draws of a
    # 9-spin model come back as the 512 distinct configurations with their counts -- no K x n download, no host pass
    m = _mvt_model(golden)
    N = 100_000_000
    with gml.Problem(model=m, num_samples=N, seed=4, histogram=True) as p:
        assert (p.K, p.n, p.M) == (512, 9, float(N))
        states, counts = p.spins(), p.counts()
        out, kkt, st = p.learn("RISE", 0.2, tol=1e-10)
    assert counts.sum() == N and len({tuple(s) for s in states}) == 512
    allstates, pr = exact_probabilities(m)
    lookup = {tuple(s): pi for s, pi in zip(allstates, pr)}
    for s, c in zip(states, counts):
        expect = lookup[tuple(s)] * N
        assert abs(c - expect) <= 6 * np.sqrt(expect) + 1
    assert np.abs(0.5 * (out + out.T) - m).max() <= 2e-3  # 1e8 samples pin the model down
    hist = gml.sample(gml.FactorGraph(m), 1_000_000, seed=4)  # the front door goes the same way
    assert hist.shape[1] == 10 and 256 <= len(hist) <= 512 and hist[:, 0].sum() == 1_000_000  # (the rarest states need more than 1e6 draws)


@pytest.mark.parametrize("form", ["RISE", "logRISE", "RPLE"])
def test_histogram_handle_learns_what_the_count_one_handle_learns(golden, form):
    m = _mvt_model(golden)
    N = 2_000_000
    with gml.Problem(model=m, num_samples=N, seed=9) as p, gml.Problem(model=m, num_samples=N, seed=9, histogram=True) as q:
        assert p.K == N and q.K <= 512 and p.M == q.M == N
        a, _, _ = p.learn(form, 0.2, tol=1e-11, precision="f64")
        b, _, sb = q.learn(form, 0.2, tol=1e-11, precision="f64")
        # the same draws: the histogram of the count-one handle's samples is the deduplicated handle
        st, ct = np.unique(p.spins(), axis=0, return_counts=True)
        qs, qc = q.spins(), q.counts()
        order = np.lexsort(qs.T[::-1])
        assert np.array_equal(qs[order], st) and np.array_equal(qc[order], ct)
    assert np.abs(a - b).max() <= 1e-9


def test_histogram_of_glauber_chains_and_limits():
    terms = {(i + 1, (i + 1) % 12 + 1): 0.3 for i in range(12)}  # a 12-spin ring
    with gml.Problem(terms=terms, n=12, num_samples=300000, seed=2, mcmc_sweeps=50, histogram=True) as p:
        assert p.K <= 4096 and p.M == 300000 and p.counts().sum() == 300000
    with pytest.raises(gml.GMLError, match="n <= 64"):
        gml.Problem(terms={(1, 70): 0.1}, n=70, num_samples=1000, histogram=True)
