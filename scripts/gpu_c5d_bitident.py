"""Config 5 at the default regulariser, full size: the solve with every eligible Hessian-vector product taken entry by entry
(gml_hv_sparse.hip) against the solve with every product as a GEMM pass -- the learned matrices must be equal bit for bit."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
L = gml._lib.lib()
L.gml_test_hv_sparse_ratio.restype = C.c_double
L.gml_test_hv_sparse_ratio.argtypes = [C.c_double]
L.gml_test_hv_sparse_calls.restype = C.c_longlong
n, K = 512, 1000000
terms = syn.block_multibody_terms(n, block=16, seed=0)
res = {}
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
    for name, ratio in (("gemm", -1.0), ("default", 0.3), ("entries", 1e30)):
        L.gml_test_hv_sparse_ratio(ratio)
        n0 = L.gml_test_hv_sparse_calls()
        t0 = time.time()
        out, kkt, st = p.learn("RISE", 0.4, tol=1e-8, precision="i8x", max_iter=150, raise_on_fail=False)
        res[name] = (out.copy(), kkt.copy(), st)
        print(name, "learn_s %.2f" % (time.time() - t0), "entry-by-entry products", L.gml_test_hv_sparse_calls() - n0,
              {k: st[k] for k in ("iterations", "passes", "hessian_passes", "hv_evals", "max_kkt", "not_converged")}, flush=True)
for name in ("default", "entries"):
    print(name, "equal to gemm bit for bit:", np.array_equal(res[name][0], res["gemm"][0]) and np.array_equal(res[name][1], res["gemm"][1]))
