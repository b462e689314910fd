"""Column compaction of the forward GEMM (csrc/gml_i8_pack.hip: k_col_union / k_build_xc; gml_dev.h: I8Pass.compact): a node tile
whose rows are sparse sweeps only the statistics columns on which one of its rows is non-zero.  The integer sums are those of the sweep
over all columns, so every result must be THE SAME BITS with the compaction on and off -- objective, gradient, and whole solves."""
import ctypes as C

import numpy as np
import pytest

import gml_amd as gml

pytestmark = pytest.mark.gpu
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
NO_COMPACT, SOLVER_COMPACT = 6, 8  # gml_solver.h: GML_TUNE_NO_COMPACT (never), GML_TUNE_SOLVER_COMPACT (gml_learn's passes too: off by default)


def tune(knob, value):
    L = gml._lib.lib()
    L.gml_test_tune.restype = C.c_double
    L.gml_test_tune.argtypes = [C.c_int, C.c_double]
    return L.gml_test_tune(knob, float(value))


@pytest.fixture
def both_ways():
    def run(fn):
        tune(NO_COMPACT, 1)
        try:
            dense = fn()
        finally:
            tune(NO_COMPACT, 0)
        tune(SOLVER_COMPACT, 1)
        try:
            return dense, fn()
        finally:
            tune(SOLVER_COMPACT, 0)
    yield run
    tune(NO_COMPACT, 0)
    tune(SOLVER_COMPACT, 0)


def sparse_rows(rng, nrows, P, nnz_lo, nnz_hi, pool=None, scale=0.3):
    th = np.zeros((nrows, P))
    for r in range(nrows):
        k = int(rng.integers(nnz_lo, nnz_hi + 1))
        cols = rng.choice(pool if pool is not None else P, size=min(k, len(pool) if pool is not None else P), replace=False)
        th[r, cols] = rng.normal(scale=scale, size=len(cols))
    return th


@pytest.mark.parametrize("form", ["RISE", "logRISE", "RPLE"])
@pytest.mark.parametrize("prec", ["i8x", "i8w"])
def test_objgrad_bits_do_not_depend_on_the_compaction(form, prec, both_ways):
    n, K = 1024, 30000  # 16 column steps: lists of up to 4 steps (256 columns) are compacted, longer ones sweep everything
    rng = np.random.default_rng(11)
    spins, _ = synthetic.block_ising(n, K, block=16, seed=3)
    counts = 1.0 + (np.arange(K) % 3)
    nodes = rng.permutation(n)[:150].astype(np.int64)  # 4 full tiles + a partial one, rows not in node order
    th = sparse_rows(rng, len(nodes), n, 3, 12, pool=np.arange(40))          # tiles 0..: a union of <= 40 columns: one step
    th[32:64] = sparse_rows(rng, 32, n, 3, 12, pool=np.arange(0, 1024, 6))  # tile 1: a union of ~150 columns: three steps
    th[64:96] = sparse_rows(rng, 32, n, 20, 60)                              # tile 2: > 256 columns in the union: not compacted
    th[96:128] = 0.0                                                        # tile 3: all rows zero: no column at all
    th[100, 1023] = 0.7                                                      # ... but one entry, in the last column
    with gml.Problem(counts=counts, spins=spins) as p:
        (f0, g0), (f1, g1) = both_ways(lambda: p.objgrad(form, nodes, th, precision=prec))
    assert np.array_equal(g0, g1)
    if form == "RPLE":  # (its f is a floating-point sum added with atomics)
        assert np.abs(f0 / f1 - 1).max() <= 1e-13
    else:
        assert np.array_equal(f0, f1)
    assert np.isfinite(g1).all() and np.abs(g1).max() > 0


def test_objgrad_multibody_bits_do_not_depend_on_the_compaction(both_ways):
    n, K = 40, 20000  # order 3: 40 + 780 statistics columns = 13 steps
    terms = synthetic.block_multibody_terms(n, block=8, seed=2)
    rng = np.random.default_rng(5)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=1, order=3) as p:
        nodes = np.arange(n, dtype=np.int64)
        th = sparse_rows(rng, n, p.P, 2, 25)
        th[:, 0] = rng.normal(scale=0.1, size=n)  # the fields: the constant column, which is not a GEMM column
        for prec in ("i8x", "i8w"):
            (f0, g0), (f1, g1) = both_ways(lambda: p.objgrad("RISE", nodes, th, precision=prec))
            assert np.array_equal(f0, f1) and np.array_equal(g0, g1)


@pytest.mark.parametrize("form,c,prec", [("RISE", 0.4, "i8x"), ("RISE", 0.4, "i8w"), ("logRISE", 0.8, "i8x"), ("RPLE", 0.2, "i8w")])
def test_learn_is_the_same_solve_with_and_without_the_compaction(form, c, prec, both_ways):
    n, K = 384, 60000
    J = synthetic.block_ising_model(n, block=16, seed=8)
    with gml.Problem(model=J, num_samples=K, seed=2) as p:
        (o0, k0, s0), (o1, k1, s1) = both_ways(lambda: p.learn(form, c, tol=1e-9, precision=prec))
    assert s0["not_converged"] == 0 and s1["not_converged"] == 0
    if form == "RPLE":  # the line search compares objective values that carry atomics' rounding: same optimum, maybe another path
        assert np.abs(o0 - o1).max() <= 2e-9
    else:
        assert np.array_equal(o0, o1) and np.array_equal(k0, k1)
        assert (s0["iterations"], s0["passes"], s0["forward_passes"]) == (s1["iterations"], s1["passes"], s1["forward_passes"])


def test_learn_order3_same_solve_with_and_without_the_compaction(both_ways):
    n, K = 36, 40000
    terms = synthetic.block_multibody_terms(n, block=12, seed=1)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=2, order=3) as p:
        (o0, k0, s0), (o1, k1, s1) = both_ways(lambda: p.learn("RISE", 0.6, tol=1e-9, precision="i8x"))
    assert np.array_equal(o0, o1) and s0["iterations"] == s1["iterations"] and s1["not_converged"] == 0
