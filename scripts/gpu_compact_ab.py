"""Column compaction of the forward GEMM (round 6): learn() wall-clock and pass time with the compaction off / on, interleaved on one
box (gml_test_tune knob 6 = never compact; knob 7 = the timing hook compacts too; by default operator calls compact, and gml_learn's passes do from 4 096 statistics columns on).  Solutions are compared bit for bit."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gml_amd as gml  # noqa: E402

syn = __import__("importlib").import_module("gml_amd.synthetic")
L = gml._lib.lib()
L.gml_test_tune.restype = C.c_double
L.gml_test_tune.argtypes = [C.c_int, C.c_double]


def learn_ab(tag, p, form, c, prec, tol=1e-9, reps=3, **kw):
    res = {}
    for rep in range(reps):
        for mode in ("dense", "compact"):
            L.gml_test_tune(6, 1.0 if mode == "dense" else 0.0)  # dense: never compact; compact: the library's default rule (gml_learn
            # tries from 4 096 statistics columns on, with a back-off after passes that came out dense)
            t0 = time.perf_counter()
            out, kkt, st = p.learn(form, c, tol=tol, precision=prec, raise_on_fail=False, **kw)
            res.setdefault(mode, []).append((time.perf_counter() - t0, out, st))
    L.gml_test_tune(6, 0.0)
    td, tc = sorted(t for t, _, _ in res["dense"])[reps // 2], sorted(t for t, _, _ in res["compact"])[reps // 2]
    same = np.array_equal(res["dense"][-1][1], res["compact"][-1][1])
    sd, sc = res["dense"][-1][2], res["compact"][-1][2]
    print(f"{tag:<44s} {prec}: learn() all columns {td * 1e3:9.2f} ms (t_pass {sd['t_pass'] * 1e3:8.2f})   default rule {tc * 1e3:9.2f} ms "
          f"(t_pass {sc['t_pass'] * 1e3:8.2f})   x{td / tc:.2f}   iterations {sd['iterations']}/{sc['iterations']}  same bits: {same}", flush=True)


def pass_ab(tag, p, form, theta, prec):
    out = {}
    for rep in range(2):
        for mode in ("dense", "compact"):
            L.gml_test_tune(7, 0.0 if mode == "dense" else 1.0)
            km = p.bench_pass_resident(form, theta, steps=20, warmup=3, precision=prec)
            out[mode] = km
    L.gml_test_tune(7, 0.0)
    d, c = out["dense"], out["compact"]
    print(f"{tag:<44s} {prec}: pass all columns {d['device_ms_per_pass']:.3f} ms (fwd {d['fwd_ms']:.3f} bwd {d['bwd_ms']:.3f})   compacted "
          f"{c['device_ms_per_pass']:.3f} ms (fwd {c['fwd_ms']:.3f} bwd {c['bwd_ms']:.3f})", flush=True)


which = sys.argv[1:] or ["headline", "c3", "c4", "c2", "c5"]
if "headline" in which or "c3" in which:
    J = syn.block_ising_model(1024, block=16, seed=0)
    with gml.Problem(model=J, num_samples=1000000, seed=0) as p:
        p.learn("RISE", 0.4, tol=1e-9, precision="i8w")
        if "headline" in which:
            for prec in ("i8w", "i8x"):
                learn_ab("headline n=1024 K=1e6 RISE(0.4)", p, "RISE", 0.4, prec)
                pass_ab("headline pass at the generating couplings", p, "RISE", J, prec)
        if "c3" in which:
            learn_ab("C3 n=1024 K=1e6 logRISE(0.8)", p, "logRISE", 0.8, "i8x")
            learn_ab("   RPLE(0.2) (denser optimum)", p, "RPLE", 0.2, "i8x")
    if "headline" in which:
        with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, 128)) as p:
            p.learn("RISE", 0.4, tol=1e-9, precision="i8w")
            learn_ab("headline, 128-node shard", p, "RISE", 0.4, "i8w", reps=5)
            learn_ab("headline, 128-node shard", p, "RISE", 0.4, "i8x", reps=5)
if "c4" in which:
    J4 = syn.block_ising_model(4096, block=8, seed=1)
    with gml.Problem(model=J4, num_samples=1000000, seed=4, node_range=(0, 512)) as p:
        p.learn("RISE", 0.4, tol=1e-9, precision="i8x")
        learn_ab("C4 shard n=4096 K=1e6 nodes 0..511 RISE(0.4)", p, "RISE", 0.4, "i8x")
        pass_ab("C4 shard pass at the generating couplings", p, "RISE", np.ascontiguousarray(J4[:512]), "i8x")
if "c2" in which:
    J2 = syn.block_ising_model(256, block=16, seed=0)
    with gml.Problem(model=J2, num_samples=100000, seed=0) as p:
        p.learn("RISE", 0.4, tol=1e-9, precision="i8x")
        learn_ab("C2 n=256 K=1e5 RISE(0.4)", p, "RISE", 0.4, "i8x", reps=5)
if "c5" in which:
    terms = syn.block_multibody_terms(512, block=16, seed=0)
    with gml.Problem(terms=terms, n=512, num_samples=1000000, seed=5, order=3) as p:
        learn_ab("C5 n=512 order 3 K=1e6 multiRISE(1.2)", p, "RISE", 1.2, "i8x", tol=1e-8, reps=1, max_iter=60)
