"""The batched Newton solve of the direction phase (k_newton_chol: A d = -pg on ragged symmetric positive definite blocks, one
workgroup per row, blocked right-looking Cholesky with panels of 32) against numpy: one panel, partial last panels, sizes around
every multiple of 32 and 64 up to 512, mixed sizes in one launch; with and without the logRISE rank-one term."""
import ctypes as C

import numpy as np
import pytest

from gml_amd import _lib

pytestmark = pytest.mark.gpu


def solve(blocks, pg, s2=0.0, g=None):
    L = _lib.lib()
    L.gml_test_newton_solve.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    R, cap = blocks.shape[0], blocks.shape[1]
    ms = np.array([b for b in pg[1]], dtype=np.int32)
    out = np.zeros((R, cap))
    _lib.check(L.gml_test_newton_solve(R, _lib._ptr(ms), cap, _lib._ptr(np.ascontiguousarray(blocks)), _lib._ptr(np.ascontiguousarray(pg[0])),
                                       float(s2), _lib._ptr(g), _lib._ptr(out), 0))
    return out


@pytest.mark.parametrize("sizes", [[1, 5, 31, 32, 33, 63, 64], [65, 90, 96, 127, 128], [129, 160, 191, 192, 193, 200, 224, 255, 256],
                                   [257, 300, 384, 511, 512], [7, 64, 100, 128, 150, 200, 256, 300, 512]])
@pytest.mark.parametrize("logrise", [False, True])
def test_batched_newton_solve_matches_numpy(sizes, logrise):
    rng = np.random.default_rng(sum(sizes))
    cap = 512
    R = len(sizes)
    blocks = np.zeros((R, cap, cap))
    pg = np.zeros((R, cap))
    g = np.zeros((R, cap)) if logrise else None
    want = []
    for r, m in enumerate(sizes):
        X = rng.choice([-1.0, 1.0], size=(4 * m + 50, m))  # a Hessian-like block: weighted +-1 outer products
        h = rng.random(4 * m + 50)
        A = (X * h[:, None]).T @ X / len(h)
        gg = rng.normal(size=m) * 0.1 if logrise else np.zeros(m)
        blocks[r, :m, :m] = A + (np.outer(gg, gg) if logrise else 0.0)  # so that A_eff = H - g g^T stays positive definite
        pg[r, :m] = rng.normal(size=m)
        if logrise:
            g[r, :m] = gg
        want.append(np.linalg.solve(A, -pg[r, :m]))
    got = solve(blocks, (pg, sizes), s2=1.0 if logrise else 0.0, g=g)
    for r, m in enumerate(sizes):
        scale = np.abs(want[r]).max()
        assert np.abs(got[r, :m] - want[r]).max() <= 1e-9 * scale, (m, np.abs(got[r, :m] - want[r]).max() / scale)


@pytest.mark.parametrize("T", [64, 128])
@pytest.mark.parametrize("logrise", [False, True])
def test_tile_preconditioner_matches_numpy(T, logrise):
    """The block-diagonal preconditioner of the matrix-free rows (launch_tile_inverse + launch_tile_apply): full and partial tiles, a
    tile of one entry, and a tile that is not positive definite (two identical statistics), which must fall back to its diagonal."""
    L = _lib.lib()
    L.gml_test_tile_precond.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(T + logrise)
    sizes = [T, T, T - 1, T // 2 + 3, 1, 17, T]
    nt = len(sizes)
    tiles = np.zeros((nt, T, T))
    g = np.zeros((nt, T))
    r = rng.normal(size=(nt, T))
    s1, s2 = 0.7, (1.0 if logrise else 0.0)
    want = []
    for t, m in enumerate(sizes):
        X = rng.choice([-1.0, 1.0], size=(6 * m + 40, m))
        if t == nt - 1:
            X[:, 5] = X[:, 9]  # duplicate statistic: singular block
        h = rng.random(len(X))
        H = (X * h[:, None]).T @ X / len(h)
        gg = rng.normal(size=m) * 0.05 if logrise else np.zeros(m)
        tiles[t, :m, :m] = H / s1 + (np.outer(gg, gg) / s1 if logrise else 0.0)  # so that s1 H - s2 g g^T = the SPD matrix above
        tiles[t, m:, m:] = np.eye(T - m) * 3.0  # padding entries (the kernels must not touch them)
        g[t, :m] = gg
        A = s1 * tiles[t, :m, :m] - s2 * np.outer(gg, gg)
        want.append(r[t, :m] / np.diag(A) if t == nt - 1 else np.linalg.solve(A, r[t, :m]))
    z = np.zeros((nt, T))
    ms = np.array(sizes, dtype=np.int32)
    _lib.check(L.gml_test_tile_precond(T, nt, _lib._ptr(ms), _lib._ptr(np.ascontiguousarray(tiles)), s1, s2, _lib._ptr(g), _lib._ptr(r), _lib._ptr(z), 0))
    for t, m in enumerate(sizes):
        if t == nt - 1:  # ridge-regularised inverse or the diagonal: either way a finite, symmetric positive definite action
            assert np.isfinite(z[t]).all() and r[t, :m] @ z[t, :m] > 0
            continue
        scale = np.abs(want[t]).max()
        assert np.abs(z[t, :m] - want[t]).max() <= 1e-9 * scale, (t, m, np.abs(z[t, :m] - want[t]).max() / scale)
        assert (z[t, m:] == 0).all()


@pytest.mark.parametrize("logrise", [False, True])
def test_newton_resolve_with_fixed_entries_matches_numpy(logrise):
    """The orthant-face re-solve of the Cholesky rows (k_newton_chol with fix / dfix): fixed entries keep their prescribed step, the
    others solve A_ff d_f = -pg_f - A_fx dfix_x."""
    L = _lib.lib()
    L.gml_test_newton_solve_fixed.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int]
    rng = np.random.default_rng(5 + logrise)
    sizes = [3, 33, 64, 100, 190, 257, 512]
    cap, R = 512, len(sizes)
    blocks = np.zeros((R, cap, cap)); pg = np.zeros((R, cap)); g = np.zeros((R, cap)) if logrise else None
    fix = np.zeros((R, cap), dtype=np.uint8); dfix = np.zeros((R, cap))
    want = []
    for r, m in enumerate(sizes):
        X = rng.choice([-1.0, 1.0], size=(4 * m + 50, m)); h = rng.random(len(X))
        A = (X * h[:, None]).T @ X / len(h)
        gg = rng.normal(size=m) * 0.1 if logrise else np.zeros(m)
        blocks[r, :m, :m] = A + np.outer(gg, gg)
        pg[r, :m] = rng.normal(size=m)
        if logrise:
            g[r, :m] = gg
        fx = rng.random(m) < 0.2
        fx[0] = True  # (at least one fixed, and one with a non-zero step)
        fix[r, :m] = fx
        dfix[r, :m] = np.where(fx, np.where(rng.random(m) < 0.5, 0.0, rng.normal(size=m) * 0.01), 0.0)
        dfix[r, 0] = 0.02
        d = dfix[r, :m].copy()
        fr = ~fx
        if fr.any():
            d[fr] = np.linalg.solve(A[np.ix_(fr, fr)], -pg[r, :m][fr] - A[np.ix_(fr, fx)] @ dfix[r, :m][fx])
        want.append(d)
    out = np.zeros((R, cap))
    ms = np.array(sizes, dtype=np.int32)
    _lib.check(L.gml_test_newton_solve_fixed(R, _lib._ptr(ms), cap, _lib._ptr(blocks), _lib._ptr(pg), 1.0 if logrise else 0.0, _lib._ptr(g),
                                             _lib._ptr(fix), _lib._ptr(dfix), _lib._ptr(out), 0))
    for r, m in enumerate(sizes):
        scale = np.abs(want[r]).max()
        assert np.abs(out[r, :m] - want[r]).max() <= 1e-9 * scale, (m, np.abs(out[r, :m] - want[r]).max() / scale)


# ---- gml_learn_warm: the same optimum from any starting point, in fewer iterations from a near one ----------------------------------
@pytest.mark.parametrize("form,c", [("RISE", 0.4), ("logRISE", 0.8), ("RPLE", 0.2)])
@pytest.mark.parametrize("prec", ["i8x", "i8w", "f64"])
def test_warm_start_reaches_the_cold_optimum(form, c, prec):
    import gml_amd as gml
    synthetic = __import__("importlib").import_module("gml_amd.synthetic")
    n, K = 96, 40000
    J = synthetic.block_ising_model(n, block=16, seed=4)
    with gml.Problem(model=J, num_samples=K, seed=1) as p:
        cold, kc, sc = p.learn(form, c, tol=1e-10, precision=prec)
        # from the optimum itself: the certifying pass, at most a polishing step
        again, ka, sa = p.learn(form, c, tol=1e-10, precision=prec, x0=cold)
        # (i8x at this tolerance ends on the FP64 path, below the noise floor of its own gradient: its restart re-enters that phase)
        assert sa["iterations"] <= (2 if prec != "i8x" else sc["iterations"]) and ka.max() <= 1e-10 and np.abs(again - cold).max() <= 2e-9
        # a regularisation path: 2c -> c from the previous solution
        far, _, sf = p.learn(form, 2 * c, tol=1e-10, precision=prec)
        warm, kw, sw = p.learn(form, c, tol=1e-10, precision=prec, x0=far)
        assert kw.max() <= 1e-10 and np.abs(warm - cold).max() <= 2e-9 and sw["iterations"] <= sc["iterations"]
        assert ((warm != 0) == (cold != 0)).all()
        # from a dense random point (every coordinate starts off its optimum, most of them off zero)
        rnd = np.random.default_rng(0).normal(scale=0.05, size=cold.shape)
        w2, k2, s2 = p.learn(form, c, tol=1e-10, precision=prec, x0=rnd)
        assert k2.max() <= 1e-10 and np.abs(w2 - cold).max() <= 2e-9
        bad = cold.copy()
        bad[3, 7] = np.nan
        with pytest.raises(gml.GMLError, match="non-finite"):
            p.learn(form, c, tol=1e-10, precision=prec, x0=bad)
        with pytest.raises(gml.GMLError, match="shape"):
            p.learn(form, c, tol=1e-10, precision=prec, x0=cold[:5])


def test_warm_start_multibody_and_node_shard():
    import gml_amd as gml
    synthetic = __import__("importlib").import_module("gml_amd.synthetic")
    n, K = 24, 30000
    terms = synthetic.block_multibody_terms(n, block=12, seed=7)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=2, order=3, node_range=(5, 17)) as p:
        cold, kc, sc = p.learn("RISE", 0.5, tol=1e-10)
        far, _, _ = p.learn("RISE", 1.0, tol=1e-6)
        warm, kw, sw = p.learn("RISE", 0.5, tol=1e-10, x0=far)
        assert cold.shape == (12, p.P) and kw.max() <= 1e-10 and np.abs(warm - cold).max() <= 2e-9
        again, ka, sa = p.learn("RISE", 0.5, tol=1e-10, x0=cold)
        assert sa["iterations"] <= 2 and np.abs(again - cold).max() <= 2e-9
