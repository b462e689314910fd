"""Config 5 (order 3, n=512, 131 k columns, 1e6 samples) at the sparse regulariser with precision i8w: the wide recombination
path of the FP64-grade forward kernel inside learn()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gml_amd as gml
from oracle import oracle as O
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
n, K = 512, 1000000
terms = synthetic.block_multibody_terms(n, block=16, seed=0)
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
    for prec in ("i8x", "i8w"):
        for rep in range(2):
            t0 = time.perf_counter(); out, kkt, st = p.learn("RISE", 1.2, tol=1e-8, precision=prec, max_iter=60); t = time.perf_counter() - t0
        print(prec, f"{t:.2f} s", {k: st[k] for k in ("iterations", "passes", "forward_passes", "max_kkt", "not_converged", "t_pass", "t_hess")}, flush=True)
        if prec == "i8x": ox = out
    print("max |i8w - i8x|", np.abs(out - ox).max())
    some = np.array([0, 511])
    fw, gw = p.objgrad("RISE", some, out[some], precision="i8w")
    fx, gx = p.objgrad("RISE", some, out[some], precision="i8x")
    spins = p.spins()
fo, go = O.objgrad_multi3_nodes(None, spins, some, out[some])
print("i8w vs oracle", np.abs(fw / fo - 1).max(), np.abs(gw - go).max(), " i8x vs oracle", np.abs(fx / fo - 1).max(), np.abs(gx - go).max())
