"""BASELINE.json configs 3, 4 and 5 and the headline config at FULL size on the MI355X, through the C ABI, checked against the
CPU oracle's blocked restatement (oracle/gml_oracle_fast.c).  The oracle cannot solve every node at these sizes in seconds, so
for the pairwise configs 40 nodes are SOLVED by it -- 32 drawn at random and the 8 the library itself reports the largest KKT
residual for -- and the north-star metric is asserted on them directly (/root/reference/test/runtests.jl:66-102 compares learned
matrices with stored ones the same way):
    ||x - x_ref|| / ||x_ref|| <= 1e-6,   max|x - x_ref| / max|x_ref| <= 1e-6,   the same zero pattern,
next to the solver-independent KKT residual of the library's rows evaluated with the ORACLE's gradient on the same 40 nodes and the
oracle's objective/gradient at a point off the optimum.  The order-3 config has no oracle solver at this size (130 817
parameters per node): its rows are certified by the oracle's order-3 gradient (KKT) on 12 nodes.  Samples are drawn on the
device (no multi-GB host matrix is built) and downloaded once for the oracle.

Wall-clock guards are 3x the measured solve (a second solve on the warm handle: the first one of a process also pays for the
allocation of its workspace), so that a performance regression fails this suite.
"""
import time

import numpy as np
import pytest

import gml_amd as gml
from oracle import oracle as O

pytestmark = pytest.mark.gpu
synthetic = __import__("importlib").import_module("gml_amd.synthetic")


def _oracle_kkt(form, spins, rows, nodes, lam):
    f, g = O.objgrad_nodes(form, None, spins, np.asarray(nodes), rows)
    return max(O.kkt_residual(rows[a], g[a], lam, int(u)) for a, u in enumerate(nodes))


def _checked_nodes(kkt, node0, seed):
    """32 random local nodes + the 8 with the largest KKT residual the library reported (global node ids)"""
    rng = np.random.default_rng(seed)
    R = len(kkt)
    worst = np.argsort(kkt)[-8:]
    rest = np.setdiff1d(np.arange(R), worst)
    pick = np.concatenate([rng.choice(rest, size=min(32, len(rest)), replace=False), worst])
    return np.sort(pick) + node0


def _assert_solution_parity(form, c, spins, out, kkt, node0, lam, seed, kkt_tol=5e-9):
    """the library's rows against the oracle's own SOLUTIONS of the same node problems (north-star: 1e-6 relative), and their KKT
    residual by the oracle's gradient"""
    nodes = _checked_nodes(kkt, node0, seed)
    rows = out[nodes - node0]
    ref, rk, _ = O.learn_nodes_fast(None, spins, nodes, form, c=c, tol=1e-10)
    assert rk.max() <= 1e-9
    rel_fro = np.linalg.norm(rows - ref) / np.linalg.norm(ref)
    rel_max = np.abs(rows - ref).max() / np.abs(ref).max()
    assert rel_fro <= 1e-6 and rel_max <= 1e-6, (rel_fro, rel_max)
    assert ((rows == 0) == (ref == 0)).all()
    assert _oracle_kkt(form, spins, rows, nodes, lam) <= kkt_tol
    return rel_fro, rel_max


def _kkt_all_nodes_by_wide_limbs(p, form, out, node0, lam):
    """KKT residual of EVERY learned row from the library's FP64-grade gradient (precision i8w: held to 1e-12 against the oracle by
    tests/test_gpu_parity.py, and an arithmetic independent of the 38/31-bit one that produced an i8x solution): the whole-problem
    certificate next to the oracle's on the 40 sampled nodes."""
    R, n = out.shape
    nodes = np.arange(node0, node0 + R)
    _, g = p.objgrad(form, nodes, out, precision="i8w")
    pg = np.where(out > 0, g + lam, np.where(out < 0, g - lam, np.sign(g) * np.maximum(np.abs(g) - lam, 0)))
    pg[np.arange(R), nodes] = g[np.arange(R), nodes]  # the field slot is not penalised (:171)
    return float(np.abs(pg).max())


def _timed_learn(p, *args, **kw):
    """(result of the first solve, wall-clock of a second solve on the warm handle)"""
    res = p.learn(*args, **kw)
    t0 = time.perf_counter()
    p.learn(*args, **kw)
    return res, time.perf_counter() - t0


@pytest.mark.parametrize("prec", ["i8x", "i8w"])
def test_c3_logrise_full_size(prec):
    # config 3: n=1024 random (16-spin block) Ising, 1e6 samples, logRISE(0.8) with l1, one MI355X
    n, K = 1024, 1000000
    J = synthetic.block_ising_model(n, block=16, seed=0)
    some = np.array([0, 333, 640, 1023])
    with gml.Problem(model=J, num_samples=K, seed=3) as p:
        (out, kkt, st), t_learn = _timed_learn(p, "logRISE", 0.8, tol=1e-9, precision=prec)
        lam = st["lambda_"]
        rng = np.random.default_rng(0)
        th = out[some] + rng.normal(scale=0.02, size=(4, n)) * (rng.random((4, n)) < 0.05)  # off the optimum
        f8, g8 = p.objgrad("logRISE", some, th, precision=prec)
        f64, g64 = p.objgrad("logRISE", some, th, precision="f64")
        kkt_all = _kkt_all_nodes_by_wide_limbs(p, "logRISE", out, 0, lam)
        spins = p.spins()
    assert st["not_converged"] == 0 and kkt.max() <= 1e-9 and st["polished"] == 0
    assert kkt_all <= (3e-9 if prec == "i8x" else 1.0001e-9)  # all 1024 rows, by a gradient the solve did not use (i8x) / at 1e-12 (i8w)
    assert t_learn < (0.35 if prec == "i8x" else 0.55)  # measured 0.115 s (i8x), 0.17 s (i8w)
    fo, go = O.objgrad_nodes("logRISE", None, spins, some, th)
    assert np.abs(f64 - fo).max() <= 1e-12 and np.abs(g64 - go).max() <= 1e-12      # FP64 path = the oracle
    ptol = 1e-8 if prec == "i8x" else 1e-12
    assert np.abs(f8 - fo).max() <= ptol and np.abs(g8 - go).max() <= ptol          # int8-limb paths
    _assert_solution_parity("logRISE", 0.8, spins, out, kkt, 0, lam, seed=3)         # the learned rows are the oracle's
    sym = 0.5 * (out + out.T)
    assert np.abs(sym - J).max() <= 0.06                                             # and the generating model


@pytest.mark.parametrize("node_range", [(0, 512), (3584, 4096)])
def test_c4_sparse_ising_shard_full_size(node_range):
    # config 4: n=4096 sparse (8-spin block) Ising, 1e6 samples, RISE(); the shard one rank of 8 owns
    n, K = 4096, 1000000
    J = synthetic.block_ising_model(n, block=8, seed=1)
    n0, n1 = node_range
    some = np.array([n0, n0 + 77, n0 + 300, n1 - 1])
    with gml.Problem(model=J, num_samples=K, seed=4, node_range=node_range) as p:
        (out, kkt, st), t_learn = _timed_learn(p, "RISE", 0.4, tol=1e-9, precision="i8x")
        lam = st["lambda_"]
        f8, g8 = p.objgrad("RISE", some, J[some], precision="i8x")
        fw, gw = p.objgrad("RISE", some, J[some], precision="i8w")
        kkt_all = _kkt_all_nodes_by_wide_limbs(p, "RISE", out, n0, lam)
        spins = p.spins()
    assert out.shape == (512, n) and st["not_converged"] == 0 and kkt.max() <= 1e-9 and kkt_all <= 3e-9
    assert t_learn < 0.8  # measured 0.255 s
    fo, go = O.objgrad_nodes("RISE", None, spins, some, J[some])
    assert np.abs(f8 / fo - 1).max() <= 1e-8 and np.abs(g8 - go).max() <= 1e-8
    assert np.abs(fw / fo - 1).max() <= 1e-12 and np.abs(gw - go).max() <= 1e-12
    _assert_solution_parity("RISE", 0.4, spins, out, kkt, n0, lam, seed=4)
    assert np.abs(out[:, n0:n1] - J[n0:n1, n0:n1]).max() <= 0.06  # un-symmetrised rows vs the generating model
    off = out.copy()
    off[:, n0:n1] = 0
    assert np.abs(off).max() <= 0.02                               # nothing outside the diagonal blocks


def test_c4_whole_problem_dress_rehearsal_on_one_gpu():
    # config 4 end to end, everything but distinct devices: the K x (1 + n) histogram a caller holds (int8 here: 4.1 GB; the
    # reference's Matrix{Int64} is 32.8 GB) -> gml_multi_create with EIGHT parts (packed once on the host, the bits copied per part;
    # all parts on device 0) -> gml_multi_learn with dev_out (every part's rows written device to device into its block, the gather
    # in place) -> 0.5 (R + R^T) as learn() does (:184-188).  What the first 8-GPU run adds is eight different device ids.
    import torch
    n, K, parts = 4096, 1000000, 8
    J = synthetic.block_ising_model(n, block=8, seed=1)
    shards = {}
    with gml.Problem(model=J, num_samples=K, seed=4, node_range=(0, 512)) as p:
        shards[0] = p.learn("RISE", 0.4, tol=1e-9, precision="i8x")[0]
        hist = np.empty((K, n + 1), dtype=np.int8)
        hist[:, 0] = 1
        hist[:, 1:] = p.spins()
    with gml.Problem(hist, node_range=(3584, 4096)) as p:  # (the other end of the node range, from the caller's matrix)
        shards[7] = p.learn("RISE", 0.4, tol=1e-9, precision="i8x")[0]
    bufs = [torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda:0") for _ in range(parts)]
    t0 = time.perf_counter()
    with gml.MultiProblem(hist, [0] * parts) as m:
        t_create = time.perf_counter() - t0
        assert (m.n, m.K, m.P, m.ndev) == (n, K, n, parts)
        t0 = time.perf_counter()
        out, kkt, st = m.learn("RISE", 0.4, tol=1e-9, precision="i8x", dev_out=[b.data_ptr() for b in bufs])
        t_learn = time.perf_counter() - t0
        stats = m.part_stats()
        kind = m.gather_kind()
    torch.cuda.synchronize()
    print(f"C4 dress rehearsal: create {t_create:.2f} s, learn + gather {t_learn:.2f} s ({kind}), per-part solves "
          f"{[round(q['t_total'], 2) for q in stats]} s, iterations {[q['iterations'] for q in stats]}")
    assert st["not_converged"] == 0 and kkt.max() <= 1e-9 and out.shape == (n, n)
    assert len(stats) == parts and all(q["iterations"] > 0 and q["not_converged"] == 0 for q in stats)
    assert sum(q["node_evals"] for q in stats) == st["node_evals"]
    # the rows of a part do not depend on the other parts, nor on how the handle came about: bit for bit the single-shard solves
    assert np.array_equal(out[0:512], shards[0]) and np.array_equal(out[3584:4096], shards[7])
    assert kind == "peer-copy"  # (a device list that repeats a GPU; distinct devices: rccl-allgather)
    for b in (bufs[0], bufs[parts - 1]):  # the gathered matrix is on every part's device
        assert np.array_equal(b.cpu().numpy(), out)
    sym = 0.5 * (out + out.T)
    assert np.abs(sym - J).max() <= 0.06
    assert t_create < 1.0 and t_learn < 5.0  # measured 0.13 s and 1.67 s (eight parts' solves sharing one GPU: 1.2 - 1.65 s each)


def _multi3_kkt(spins, out, nodes, lam):
    """KKT residual of order-3 rows by the ORACLE's order-3 gradient; slot 0 (the field, key (u,)) is not penalised (:118)"""
    worst = 0.0
    for a0 in range(0, len(nodes), 4):  # (four nodes per call: the oracle holds their statistics columns at once)
        nd = np.asarray(nodes[a0:a0 + 4])
        fo, go = O.objgrad_multi3_nodes(None, spins, nd, out[nd])
        for a, u in enumerate(nd):
            x, g = out[u], go[a]
            pg = np.where(x > 0, g + lam, np.where(x < 0, g - lam, np.sign(g) * np.maximum(np.abs(g) - lam, 0)))
            pg[0] = g[0]
            worst = max(worst, float(np.abs(pg).max()))
    return worst


@pytest.mark.parametrize("prec", ["i8x", "i8w"])
def test_c5_multibody_order3_full_size(prec):
    # config 5: n=512 spins with 3-spin interactions, 1e6 samples, multiRISE/ISODUS order 3 (130 817 parameters
    # per node, 131 k statistics columns).  Objective/gradient against the oracle's order-3 restatement on 2 nodes;
    # learn() at the regulariser of the timing runs (c = 1.2: lambda from n^2 as in :86, sparse optimum).
    n, K = 512, 1000000
    terms = synthetic.block_multibody_terms(n, block=16, seed=0)
    some = np.array([0, 511])
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
        P = p.P
        assert P == 1 + 511 + 511 * 510 // 2
        (out, kkt, st), t_learn = _timed_learn(p, "RISE", 1.2, tol=1e-8, precision=prec, max_iter=60)
        lam = st["lambda_"]
        keys0 = p.multi_keys(0)
        th = out[some].copy()
        f8, g8 = p.objgrad("RISE", some, th, precision=prec)
        spins = p.spins()
    assert st["not_converged"] == 0 and kkt.max() <= 1e-8
    assert keys0[:3] == [(0,), (0, 1), (0, 2)] and keys0[512] == (0, 1, 2) and len(keys0) == P
    fo, go = O.objgrad_multi3_nodes(None, spins, some, th)
    assert np.abs(f8 / fo - 1).max() <= 1e-8 and np.abs(g8 - go).max() <= 1e-8
    assert t_learn < 12.0  # measured 3.8 s
    # KKT certificate from the oracle's order-3 gradient on 12 nodes: 8 at random + the 4 with the largest reported residual
    # (multiRISE at order >= 3 has no golden in the reference and no oracle SOLVER at this size: parity = this certificate)
    nodes = np.unique(np.concatenate([np.random.default_rng(5).choice(n, 8, replace=False), np.argsort(kkt)[-4:]]))
    assert _multi3_kkt(spins, out, nodes, lam) <= 5e-8
    err = max(abs(v - terms.get(tuple(sorted(i + 1 for i in key)), 0.0)) for key, v in zip(keys0, out[0]))
    assert err <= 0.06  # node 0's terms are the generating ones up to sampling noise and the l1 shrinkage


def test_c5_default_regulariser_full_size():
    # config 5 AS BASELINE.json STATES IT: `ISODUS()` = multiRISE(0.4, true, 3), the reference's default regulariser (:28).
    # lambda comes from n^2, not from the 130 817 parameters per node (:86), so the optimum is DENSE: thousands of the
    # noise coefficients exceed lambda and a node's support outgrows the 512-entry Cholesky block -- those rows run the
    # matrix-free Newton-CG (Hessian-vector products on the int8 cores over a sub-sample of the configurations), the rest
    # the Cholesky blocks.  All 512 nodes must converge; the KKT certificate comes from the ORACLE's order-3 gradient.
    n, K = 512, 1000000
    terms = synthetic.block_multibody_terms(n, block=16, seed=0)
    some = np.array([0, 257])
    with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
        t0 = time.time()
        out, kkt, st = p.learn("RISE", gml.ISODUS().regularizer, tol=1e-8, precision="i8x", max_iter=120)
        t_learn = time.time() - t0
        lam = st["lambda_"]
        keys0 = p.multi_keys(0)
        spins = p.spins()
    print(f"C5 at the default regulariser: learn() {t_learn:.1f} s, {st['iterations']} iterations, {st['passes']} + {st['forward_passes']} passes, "
          f"{st['hessian_passes']} Hessian steps, {st['hv_evals']} H.v node evaluations, max support {int((out != 0).sum(1).max())}")
    assert gml.ISODUS().regularizer == 0.4 and lam == pytest.approx(0.4 * np.sqrt(np.log(n * n / 0.05) / K), rel=1e-12)
    assert st["not_converged"] == 0 and kkt.max() <= 1e-8
    supp = (out != 0).sum(1)
    assert supp.min() > 512  # every support outgrows the Cholesky block (early iterations run on blocks, the rest matrix-free)
    assert st["hv_evals"] > 0 and st["hessian_passes"] > st["iterations"]
    nodes = np.unique(np.concatenate([some, np.random.default_rng(6).choice(n, 6, replace=False), np.argsort(kkt)[-4:]]))
    assert _multi3_kkt(spins, out, nodes, lam) <= 5e-8
    err = max(abs(v - terms.get(tuple(sorted(i + 1 for i in key)), 0.0)) for key, v in zip(keys0, out[0]))
    assert err <= 0.06
    assert t_learn < 72.0  # measured 23.6 s (first solve of the handle; 34-36 s before round 4's relaxed / entry-by-entry Hessian-vector products)


@pytest.mark.parametrize("form,c,prec", [("RPLE", 0.2, "i8x"), ("RISE", 0.4, "i8x"), ("RISE", 0.4, "i8w")])
def test_headline_size_solutions_match_the_oracle(form, c, prec):
    # The headline config (n=1024, K=1e6).  RPLE at its default regulariser has the denser optimum of the three (lambda = 0.2
    # sqrt(log(n^2/0.05)/M) lies below the sampling noise 1/sqrt(M)): working sets of several hundred entries, the large-block
    # paths of the Newton solve.
    n, K = 1024, 1000000
    J = synthetic.block_ising_model(n, block=16, seed=0)
    with gml.Problem(model=J, num_samples=K, seed=3) as p:
        (out, kkt, st), t_learn = _timed_learn(p, form, c, tol=1e-9, precision=prec)
        lam = st["lambda_"]
        kkt_all = _kkt_all_nodes_by_wide_limbs(p, form, out, 0, lam)
        spins = p.spins()
    assert st["not_converged"] == 0 and kkt.max() <= 1e-9 and kkt_all <= (3e-9 if prec == "i8x" else 1.0001e-9)
    if form == "RISE":
        assert t_learn < (0.35 if prec == "i8x" else 0.55)  # measured 0.113 s (i8x), 0.17 s (i8w)
    _assert_solution_parity(form, c, spins, out, kkt, 0, lam, seed=7)
    assert np.abs(0.5 * (out + out.T) - J).max() <= 0.06
