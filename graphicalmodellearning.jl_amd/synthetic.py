"""Synthetic +-1 sample generators for the benchmark configurations.

The reference's sampler enumerates all 2^n states (sampling.jl:34-57), which stops at n ~ 25;
the benchmark models are block-diagonal so the same exact method (CDF inversion over the 2^b
states of a block) applies block-wise.  Rows are distinct with overwhelming probability, so the
histogram has one row per sample and all counts are 1.
"""
import numpy as np


def block_ising(n, K, block=16, seed=0, p_edge=0.3, jmin=0.2, jmax=0.6, hmax=0.1):
    """Returns (spins int8 [K, n], J float64 [n, n] with fields on the diagonal)."""
    assert n % block == 0
    rng = np.random.default_rng(seed)
    nb = n // block
    J = np.zeros((n, n))
    spins = np.empty((K, n), dtype=np.int8)
    states = (((np.arange(2 ** block)[:, None] >> np.arange(block)) & 1) * 2 - 1).astype(np.int8)
    sf = states.astype(np.float64)
    for B in range(nb):
        A = np.zeros((block, block))
        iu = np.triu_indices(block, 1)
        on = rng.random(len(iu[0])) < p_edge
        vals = rng.choice([-1.0, 1.0], size=len(iu[0])) * rng.uniform(jmin, jmax, size=len(iu[0]))
        A[iu] = np.where(on, vals, 0.0)
        A = A + A.T
        h = rng.uniform(-hmax, hmax, size=block)
        en = 0.5 * ((sf @ A) * sf).sum(1) + sf @ h  # weigh_proba (sampling.jl:26-30)
        pr = np.exp(en - en.max())
        cdf = np.cumsum(pr / pr.sum())
        idx = np.minimum(np.searchsorted(cdf, rng.random(K)), 2 ** block - 1)
        spins[:, B * block:(B + 1) * block] = states[idx]
        J[B * block:(B + 1) * block, B * block:(B + 1) * block] = A + np.diag(h)
    return spins, J


def block_ising_model(n, block=16, seed=0, p_edge=0.3, jmin=0.2, jmax=0.6, hmax=0.1):
    """The model only (n x n, fields on the diagonal), same family as block_ising: for sampling on the device
    (gml.Problem(model=J, num_samples=K)), where no host sample matrix is built."""
    assert n % block == 0
    rng = np.random.default_rng(seed)
    J = np.zeros((n, n))
    iu = np.triu_indices(block, 1)
    for B in range(n // block):
        on = rng.random(len(iu[0])) < p_edge
        vals = rng.choice([-1.0, 1.0], size=len(iu[0])) * rng.uniform(jmin, jmax, size=len(iu[0]))
        A = np.zeros((block, block))
        A[iu] = np.where(on, vals, 0.0)
        A = A + A.T + np.diag(rng.uniform(-hmax, hmax, size=block))
        J[B * block:(B + 1) * block, B * block:(B + 1) * block] = A
    return J


def block_multibody_terms(n, block=12, seed=0, p_edge=0.3, n_triples=None):
    """The model only, as {1-based sorted key: weight}: blocks with fields, pairwise terms and random triples
    (config C5), for sampling on the device (gml.Problem(terms=..., n=n, num_samples=K))."""
    assert n % block == 0
    rng = np.random.default_rng(seed)
    terms = {}
    n_triples = block if n_triples is None else n_triples
    for B in range(n // block):
        base = B * block
        for i in range(block):
            terms[(base + i + 1,)] = rng.uniform(-0.1, 0.1)
            for j in range(i + 1, block):
                if rng.random() < p_edge:
                    terms[(base + i + 1, base + j + 1)] = rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.6)
        for _ in range(n_triples):
            i, j, k = sorted(int(v) for v in rng.choice(block, size=3, replace=False))
            key = (base + i + 1, base + j + 1, base + k + 1)
            terms[key] = terms.get(key, 0.0) + rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.5)
    return terms


def block_multibody(n, K, block=12, seed=0, p_edge=0.3, n_triples=None):
    """Blocks with pairwise terms plus random triples (config C5).  Returns (spins, terms dict
    with 1-based sorted keys)."""
    assert n % block == 0
    rng = np.random.default_rng(seed)
    spins = np.empty((K, n), dtype=np.int8)
    states = (((np.arange(2 ** block)[:, None] >> np.arange(block)) & 1) * 2 - 1).astype(np.int8)
    sf = states.astype(np.float64)
    terms = {}
    n_triples = block if n_triples is None else n_triples
    for B in range(n // block):
        en = np.zeros(2 ** block)
        base = B * block
        for i in range(block):
            h = rng.uniform(-0.1, 0.1)
            terms[(base + i + 1,)] = h
            en += h * sf[:, i]
            for j in range(i + 1, block):
                if rng.random() < p_edge:
                    v = rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.6)
                    terms[(base + i + 1, base + j + 1)] = v
                    en += v * sf[:, i] * sf[:, j]
        for _ in range(n_triples):
            i, j, k = sorted(rng.choice(block, size=3, replace=False))
            v = rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.5)
            terms[(base + i + 1, base + j + 1, base + k + 1)] = terms.get((base + i + 1, base + j + 1, base + k + 1), 0.0) + v
            en += v * sf[:, i] * sf[:, j] * sf[:, k]  # monomial energy (sampling.jl:60-65)
        pr = np.exp(en - en.max())
        cdf = np.cumsum(pr / pr.sum())
        idx = np.minimum(np.searchsorted(cdf, rng.random(K)), 2 ** block - 1)
        spins[:, base:base + block] = states[idx]
    return spins, terms


def enumerate_sample(J, N, seed=0):
    """Exact sampling of a small pairwise model by full enumeration -- what the reference's
    `sample(gm, N)` does (sampling.jl:34-57) with numpy's RNG.  Returns the histogram matrix
    [count, s_1..s_n] (one row per observed configuration)."""
    J = np.asarray(J, dtype=float)
    n = J.shape[0]
    rng = np.random.default_rng(seed)
    states = ((np.arange(2 ** n)[:, None] >> np.arange(n)) & 1) * 2 - 1  # int_to_spin (sampling.jl:11-14)
    sf = states.astype(float)
    A = J - np.diag(np.diag(J))
    en = 0.5 * ((sf @ A) * sf).sum(1) + sf @ np.diag(J)
    pr = np.exp(en - en.max())
    pr /= pr.sum()
    counts = rng.multinomial(N, pr)
    keep = counts > 0
    return np.concatenate([counts[keep, None], states[keep]], axis=1).astype(np.int64)
