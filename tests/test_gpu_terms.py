"""Result assembly of multiRISE on the device (gml_terms_assemble / gml_learn_terms, csrc/gml_terms.hip) against the
reference's dict assembly restated in the oracle (GraphicalModelLearning.jl:129-151): bit for bit, host and device pointers,
orders 1-5; the front door learn(samples, multiRISE / ISODUS, HIP()) on a whole mid-size problem against the dict path applied
to the very same rows; and at config-5 size (n = 512, order 3, 1e6 samples), where the interpreted host loop used to take
minutes and tens of GB."""
import time

import numpy as np
import pytest

import gml_amd as gml
from oracle import oracle as O

pytestmark = pytest.mark.gpu
_lib = gml._lib
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
TermArray = __import__("importlib").import_module("gml_amd.factor_graph").TermArray


def product_row_keys(n, order, u):
    """keys of row u as the C ABI lays the row out: order 2 keeps the pairwise slots (slot i <-> spin i, slot u = field)"""
    if order == 2:
        return [(u,) if i == u else (u, i) for i in range(n)]
    return O.multi_keys(n, order, u)


def dict_assembly(rows, n, order, sym):
    rec = O.assemble_multi_dict(rows, [product_row_keys(n, order, u) for u in range(n)], sym)
    return rec, np.array([rec[k] for k in O.listing_order(rec)])


@pytest.mark.parametrize("n,order", [(1, 1), (4, 1), (2, 2), (9, 2), (3, 3), (12, 3), (36, 3), (10, 4), (9, 5)])
@pytest.mark.parametrize("sym", [True, False])
def test_assembly_kernel_equals_the_reference_dict_assembly(n, order, sym):
    P = n if order == 2 else sum(__import__("math").comb(n - 1, s - 1) for s in range(1, order + 1))
    rows = np.random.default_rng(n * 7 + order).normal(size=(n, P))
    rec, want = dict_assembly(rows, n, order, sym)
    got = _lib.terms_assemble(rows, n, order, sym)
    assert got.shape == want.shape and np.array_equal(got, want)  # same additions in the same order, one division: bit for bit
    ta = TermArray(n, order, sym, got)
    assert ta.to_dict() == rec
    # a padded leading dimension, and rows / result already on the device (what the distributed gather hands over)
    import torch
    pad = np.full((n, P + 5), np.nan)
    pad[:, :P] = rows
    assert np.array_equal(_lib.terms_assemble(pad, n, order, sym), want)
    d_rows = torch.from_numpy(pad).cuda()
    assert np.array_equal(_lib.terms_assemble(d_rows.data_ptr(), n, order, sym, ld=P + 5), want)
    d_out = torch.empty(len(want), dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().gml_terms_assemble(d_rows.data_ptr(), P + 5, n, order, int(sym), 0, d_out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_assembly_rejects_bad_arguments():
    rows = np.zeros((5, 5))
    with pytest.raises(gml.GMLError):
        _lib.terms_assemble(rows[:4], 5, 2, True)          # not all nodes
    with pytest.raises(gml.GMLError):
        _lib.terms_assemble(rows, 5, 3, True)              # rows narrower than the 11 parameters of order 3
    with pytest.raises(gml.GMLError):
        _lib.terms_assemble(rows, 5, 9, True)


@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("prec", ["i8x", "i8w", "f64"])
def test_front_door_order3_whole_problem_equals_dict_path(sym, prec, monkeypatch):
    # every node of an n = 36 order-3 problem: the FactorGraph of learn(samples, multiRISE(c, sym, 3), HIP()) -- solve and
    # assembly in one library call -- against the reference's dict assembly applied to the rows of Problem.learn, bit for bit;
    # once returned as the reference's dict (small model) and once array-backed
    n, K = 36, 40000
    terms = synthetic.block_multibody_terms(n, block=12, seed=1)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=2, order=3) as p:
        spins = p.spins()
        rows, kkt, st = p.learn("RISE", 0.6, tol=1e-9, precision=prec)
    hist = np.concatenate([np.ones((K, 1), dtype=np.int8), spins], axis=1)
    rec, want = dict_assembly(rows, n, 3, sym)
    m = gml.HIP(tol=1e-9, precision=prec)
    fg = gml.learn(hist, gml.multiRISE(0.6, sym, 3), m)
    assert isinstance(fg, gml.FactorGraph) and isinstance(fg.terms, dict) and (fg.order, fg.varible_count, fg.alphabet) == (3, n, "spin")
    # (the int8-limb passes are integer GEMMs: two solves of the same problem give the same bits; the FP64-MFMA pass accumulates
    #  with floating-point atomics, so its two solves agree to rounding only)
    same = (lambda a, b: a == b) if prec != "f64" else (lambda a, b: abs(a - b) <= 1e-13)
    assert fg.terms.keys() == rec.keys() and all(same(fg.terms[k], v) for k, v in rec.items())
    assert m.stats["not_converged"] == 0 and m.stats["t_assemble"] > 0
    monkeypatch.setattr(__import__("importlib").import_module("gml_amd.learn"), "DICT_TERMS_MAX", 0)
    fa = gml.learn(hist, gml.multiRISE(0.6, sym, 3), gml.HIP(tol=1e-9, precision=prec))
    assert isinstance(fa.terms, TermArray) and len(fa) == len(rec) and list(fa.keys()) == O.listing_order(rec)
    if prec != "f64":
        assert np.array_equal(fa.terms.weights, want) and all(fa[k] == v for k, v in rec.items())
        assert fa.jsondata() == fg.jsondata()
    else:
        assert np.abs(fa.terms.weights - want).max() <= 1e-13
    if sym:  # the generating triples come back (sampling noise + l1 shrinkage)
        assert max(abs(fa[k] - v) for k, v in terms.items() if len(k) == 3) <= 0.12


def test_front_door_order2_keeps_matching_rise():
    # runtests.jl:132-158 at a size with several node tiles: multiRISE(c, false, 2) == RISE(c, false), through both front doors
    n, K = 96, 30000
    J = synthetic.block_ising_model(n, block=16, seed=3)
    with gml.Problem(model=J, num_samples=K, seed=4) as p:
        spins = p.spins()
    hist = np.concatenate([np.ones((K, 1), dtype=np.int8), spins], axis=1)
    R = gml.learn(hist, gml.RISE(0.3, False), gml.HIP(tol=1e-10))
    two = gml.learn(hist, gml.multiRISE(0.3, False, 2), gml.HIP(tol=1e-10))
    d = gml.matrix_to_terms(R)
    assert len(two) == n * n and all(two[k] == v for k, v in d.items())  # same handle, same solve: the same bits
    sym = gml.learn(hist, gml.multiRISE(0.3, True, 2), gml.HIP(tol=1e-10))
    assert np.array_equal(sym.to_matrix(), (R + R.T) / 2.0)


def test_multi_device_route_assembles_on_the_device():
    # HIP(devices = [0, 0]): gml_multi_learn's host rows -> gml_terms_assemble
    n, K = 24, 20000
    terms = synthetic.block_multibody_terms(n, block=12, seed=5)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=6, order=3) as p:
        spins = p.spins()
    hist = np.concatenate([np.ones((K, 1), dtype=np.int8), spins], axis=1)
    one = gml.learn(hist, gml.multiRISE(0.6, True, 3), gml.HIP(tol=1e-9))
    two = gml.learn(hist, gml.multiRISE(0.6, True, 3), gml.HIP(tol=1e-9, devices=[0, 0]))
    assert one.terms == two.terms


def test_front_door_refuses_a_node_shard():
    s = np.loadtxt(__import__("os").path.join(__import__("conftest").GOLDEN, "c_samples.csv"), delimiter=",")
    with pytest.raises(ValueError, match="ALL nodes"):
        gml.learn(s, gml.multiRISE(0.2, True, 3), gml.HIP(node_range=(0, 2)))
    with gml.Problem(s, order=3, node_range=(0, 2)) as p:
        with pytest.raises(gml.GMLError, match="all nodes"):
            p.learn("RISE", 0.2, terms=True)


class PeakRSS:
    """peak resident set of this process while the block runs, over what it was at entry (sampled: psutil, 10 ms)"""

    def __enter__(self):
        import threading

        import psutil
        self._proc = psutil.Process()
        self.base = self.peak = self._proc.memory_info().rss
        self._stop = threading.Event()

        def loop():
            while not self._stop.wait(0.01):
                self.peak = max(self.peak, self._proc.memory_info().rss)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join()
        self.peak = max(self.peak, self._proc.memory_info().rss)
        self.over = self.peak - self.base


def test_c5_front_door_full_size():
    # config 5 THROUGH THE FRONT DOOR: learn(samples, ISODUS(), HIP(precision="i8x", tol=1e-8)), n = 512, 1e6 samples drawn on the
    # device, the reference's default regulariser.  What learn() adds to the solve -- the sample matrix -> handle, the assembly of
    # the 67.0 M solved parameters into 22.4 M symmetrised terms, the FactorGraph -- must stay a small fraction of it.
    n, K = 512, 1000000
    terms = synthetic.block_multibody_terms(n, block=16, seed=0)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
        spins = p.spins()
    hist = np.empty((K, n + 1), dtype=np.int8)
    hist[:, 0] = 1
    hist[:, 1:] = spins
    del spins
    m = gml.HIP(precision="i8x", tol=1e-8, max_iter=120)
    with PeakRSS() as rss:
        t0 = time.perf_counter()
        fg = gml.learn(hist, gml.ISODUS(0.4, True, 3), m)
        t_front = time.perf_counter() - t0
    st = m.stats
    t_solve = st["t_total"] - st["t_assemble"]
    overhead = t_front - t_solve
    assert isinstance(fg.terms, TermArray) and len(fg) == 512 + 512 * 511 // 2 + 512 * 511 * 510 // 6
    assert st["not_converged"] == 0 and st["max_kkt"] <= 1e-8
    # the dict path's input: the same solve into host rows (one GPU, one handle shape, integer GEMMs: the same bits as the front door's)
    with gml.Problem(hist, order=3) as p:
        t0 = time.perf_counter()
        rows, kkt_rows, st_rows = p.learn("RISE", gml.ISODUS().regularizer, tol=1e-8, precision="i8x", max_iter=120)
        t_rows = time.perf_counter() - t0
        pick = np.sort(np.random.default_rng(9).choice(n, 30, replace=False))
        slot = {int(u): {tuple(int(v) for v in k if v >= 0): j for j, k in enumerate(p.multi_keys_array(u))} for u in pick}
    print(f"C5 front door: learn() {t_front:.2f} s = solve {t_solve:.2f} s + {overhead:.2f} s (handle from the matrix {st['t_pack']:.2f} s, "
          f"assembly {st['t_assemble'] * 1e3:.0f} ms); peak host RSS over entry +{rss.over / 1e9:.2f} GB; Problem.learn into host rows "
          f"{t_rows:.2f} s; {len(fg)} terms, max support {int((rows != 0).sum(1).max())}")
    assert overhead <= 1.5 and rss.over <= 2.0e9
    # the dict path (:135-149: group by sorted key, mean in ascending u) on every key among 30 spins: 4 525 keys, bit for bit
    from itertools import combinations
    checked = 0
    for s in (1, 2, 3):
        for key in combinations([int(u) for u in pick], s):
            vals = [float(rows[u][slot[u][(u,) + tuple(v for v in key if v != u)]]) for u in key]
            assert fg[tuple(i + 1 for i in key)] == float(np.mean(vals))
            checked += 1
    assert checked == 30 + 435 + 4060
    # and the model is the generating one up to sampling noise and shrinkage
    assert max(abs(fg[k] - v) for k, v in terms.items()) <= 0.08
    nz = np.count_nonzero(fg.terms.weights)
    assert len(terms) <= nz < len(fg)
    assert t_front < 75.0  # measured 21.9 s (solve 21.8 s)


def test_sampling_from_an_array_backed_learned_model(monkeypatch):
    # runtests.jl:161-181 in spirit: learn a multi-body model, sample from the learned model.  The array-backed FactorGraph feeds the
    # device sampler its non-zero terms without a Python loop over the term list; the same draws as from the dict of those terms
    n, K = 24, 40000
    terms = synthetic.block_multibody_terms(n, block=12, seed=3)
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=4, order=3) as p:
        spins = p.spins()
    hist = np.concatenate([np.ones((K, 1), dtype=np.int8), spins], axis=1)
    monkeypatch.setattr(__import__("importlib").import_module("gml_amd.learn"), "DICT_TERMS_MAX", 0)
    fa = gml.learn(hist, gml.multiRISE(1.5, True, 3), gml.HIP(tol=1e-9))
    assert isinstance(fa.terms, TermArray)
    nz = {k: v for k, v in fa.terms.items() if v != 0.0}
    assert 0 < len(nz) < len(fa)
    s_arr = gml.sample(fa, 20000, seed=7)
    s_dict = gml.sample(gml.FactorGraph(3, n, "spin", nz), 20000, seed=7)
    assert np.array_equal(s_arr, s_dict) and s_arr[:, 0].sum() == 20000


# ---- the pairwise counterpart: 0.5 (R + R') on the device (gml_learn_matrix / gml_matrix_symmetrize, :184-186) ---------------------------
@pytest.mark.parametrize("n", [1, 3, 31, 32, 33, 100, 257])
def test_matrix_symmetrize_equals_the_host_expression(n):
    import torch
    R = np.random.default_rng(n).normal(size=(n, n))
    want = 0.5 * (R + R.T)
    assert np.array_equal(_lib.matrix_symmetrize(R), want)
    pad = np.full((n, n + 3), np.nan)
    pad[:, :n] = R
    out = np.empty((n, n))
    _lib.check(_lib.lib().gml_matrix_symmetrize(pad.ctypes.data, n + 3, n, 0, out.ctypes.data))  # a padded leading dimension
    assert np.array_equal(out, want)
    d = torch.from_numpy(R).cuda()
    _lib.check(_lib.lib().gml_matrix_symmetrize(d.data_ptr(), n, n, 0, d.data_ptr()))             # device pointers, in place
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.parametrize("form,F", [("RISE", gml.RISE), ("logRISE", gml.logRISE), ("RPLE", gml.RPLE)])
def test_pairwise_front_door_symmetrises_on_the_device(form, F):
    n, K = 160, 30000
    J = synthetic.block_ising_model(n, block=16, seed=5)
    with gml.Problem(model=J, num_samples=K, seed=6) as p:
        spins = p.spins()
        rows, _, _ = p.learn(form, F().regularizer, tol=1e-9, precision="i8w")
        sym, _, st = p.learn(form, F().regularizer, tol=1e-9, precision="i8w", matrix=True)
        raw, _, _ = p.learn(form, F().regularizer, tol=1e-9, precision="i8w", matrix=False)
    if form != "RPLE":  # (RPLE's line search compares objective values that carry atomics' rounding: two solves agree to the tolerance)
        assert np.array_equal(raw, rows) and np.array_equal(sym, 0.5 * (rows + rows.T))
    assert st["t_assemble"] > 0
    hist = np.concatenate([np.ones((K, 1), dtype=np.int8), spins], axis=1)
    got = gml.learn(hist, F(), gml.HIP(tol=1e-9, precision="i8w"))
    assert np.array_equal(got, got.T) and np.abs(got - 0.5 * (rows + rows.T)).max() <= (0 if form != "RPLE" else 2e-9)
    two = gml.learn(hist, F(), gml.HIP(tol=1e-9, precision="i8w", devices=[0, 0]))  # gathered rows -> gml_matrix_symmetrize
    assert np.array_equal(two, two.T) and np.abs(two - got).max() <= 2e-9
    with gml.Problem(hist, node_range=(0, 64)) as q:
        with pytest.raises(gml.GMLError, match="ALL nodes"):
            q.learn(form, 0.4, matrix=True)
