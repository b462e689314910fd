"""The operator boundary as an external solver sees it: gml_objgrad_batch with one row per call plays the pair the reference
registers with JuMP -- `obj(x...)` / `grad(g, x...)`, GraphicalModelLearning.jl:221-233 -- and an external first-order solver
(scipy's L-BFGS-B here; Ipopt through JuMP in the reference) drives it one node at a time, exactly like the loop :216-252.
The l1 term is handled the smooth way available to a bound-constrained solver: x = p - q, p, q >= 0, penalty
lambda * sum (p_j + q_j) over j != u -- the same epigraph trick as the reference's z >= |x| rows (:174-177)."""
import numpy as np
import pytest
from scipy.optimize import minimize

import gml_amd as gml
from conftest import load_csv
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["f64", "i8w", "auto"])  # (the Julia operator file binds auto: the FP64-grade limbs i8w, FP64 for rows they cannot hold)
@pytest.mark.parametrize("name", ["a", "c"])
def test_external_first_order_solver_reproduces_the_goldens(name, prec):
    s = load_csv(f"{name}_samples.csv")
    n = s.shape[1] - 1
    lam = O.lam(0.4, n, s[:, 0].sum())
    calls = 0
    R = np.zeros((n, n))
    with gml.Problem(s) as p:
        for u in range(n):
            pen = (np.arange(n) != u).astype(float)  # the field slot u is not penalised (:171)

            def fg(v, u=u, pen=pen):
                nonlocal calls
                calls += 1
                x = v[:n] - v[n:]
                f, g = p.objgrad("RISE", np.array([u]), x[None, :], precision=prec)  # ONE node evaluation per call
                return f[0] + lam * (pen * (v[:n] + v[n:])).sum(), np.concatenate([g[0] + lam * pen, -g[0] + lam * pen])

            bounds = [(None, None) if j == u else (0, None) for j in range(n)] + [(0, 0) if j == u else (0, None) for j in range(n)]
            res = minimize(fg, np.zeros(2 * n), jac=True, method="L-BFGS-B", bounds=bounds,
                           options=dict(ftol=1e-16, gtol=1e-12, maxiter=2000, maxcor=20))
            R[u] = res.x[:n] - res.x[n:]
    R = 0.5 * (R + R.T)  # symmetrization = true (:184-186)
    G = load_csv(f"{name}_RISE_learned.csv")
    assert calls > 3 * n
    assert np.abs(R - G).max() <= 2e-6
    assert np.linalg.norm(R - G) / np.linalg.norm(G) <= 1e-5
    R2 = gml.learn(s, gml.RISE(), gml.HIP(tol=1e-11))  # the library's own solver lands on the same matrix
    assert np.abs(R - R2).max() <= 2e-6


def test_operator_one_row_calls_equal_the_batched_call():
    # a JuMP-style callback evaluates one node per call; the batched call must give the same numbers bit for bit
    s = load_csv("mvt_samples.csv")
    n = s.shape[1] - 1
    rng = np.random.default_rng(0)
    th = rng.normal(scale=0.2, size=(n, n))
    with gml.Problem(s) as p:
        for prec in ("f64", "i8x", "i8w"):
            fb, gb = p.objgrad("RISE", np.arange(n), th, precision=prec)
            for u in range(n):
                f1, g1 = p.objgrad("RISE", np.array([u]), th[u][None, :], precision=prec)
                if prec != "f64":  # (integer arithmetic: bit for bit)
                    assert f1[0] == fb[u] and np.array_equal(g1[0], gb[u])
                else:
                    assert abs(f1[0] - fb[u]) <= 1e-14 and np.abs(g1[0] - gb[u]).max() <= 1e-14
