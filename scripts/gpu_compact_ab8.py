"""Column compaction inside gml_learn at SMALL column counts: the default rule (the solver tries from 4 096 statistics columns on) against the solver
trying at any column count (gml_test_tune knob 8), on the headline problem and its 128- / 256-node shards; interleaved medians."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import gml_amd as gml
syn = __import__("importlib").import_module("gml_amd.synthetic")
L = gml._lib.lib(); L.gml_test_tune.restype = C.c_double; L.gml_test_tune.argtypes = [C.c_int, C.c_double]
J = syn.block_ising_model(1024, block=16, seed=0)
def ab(tag, p, form, c, prec, reps=7):
    res = {"default": [], "always-try": []}
    for rep in range(reps):
        for mode in res:
            L.gml_test_tune(8, 1.0 if mode == "always-try" else 0.0)
            t0 = time.perf_counter(); out, kkt, st = p.learn(form, c, tol=1e-9, precision=prec); res[mode].append(time.perf_counter() - t0)
    L.gml_test_tune(8, 0.0)
    a, b = (sorted(v)[len(v) // 2] * 1e3 for v in res.values())
    print(f"{tag:<40s} {prec}: default {a:8.2f} ms   solver tries at any column count (back-off) {b:8.2f} ms   x{a / b:.3f}", flush=True)
for nr in ((0, 128), (0, 256), (0, 1024)):
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=nr) as p:
        p.learn("RISE", 0.4, tol=1e-9, precision="i8w")
        for prec in ("i8w", "i8x"):
            ab(f"headline nodes {nr}", p, "RISE", 0.4, prec)
        ab(f"logRISE nodes {nr}", p, "logRISE", 0.8, "i8x")
