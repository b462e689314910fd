import sys; sys.path.insert(0, '.')
import numpy as np, ctypes as C
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
L = gml._lib.lib(); L.gml_test_hv_sparse_calls.restype = C.c_longlong
L.gml_test_hv_sparse_ratio.restype = C.c_double; L.gml_test_hv_sparse_ratio.argtypes = [C.c_double]
spins, J = syn.block_ising(192, 30000, block=16, seed=7)
with gml.Problem(spins=spins) as p:
    res = {}
    for prec in ("i8x", "f64", "i8w"):
        for ratio in (-1.0, 1e30):
            L.gml_test_hv_sparse_ratio(ratio); n0 = L.gml_test_hv_sparse_calls()
            out, kkt, st = p.learn("RISE", 0.05, tol=1e-9, precision=prec, max_working=128, max_iter=200)
            res[(prec, ratio)] = out
            print(prec, ratio, "sparse calls", L.gml_test_hv_sparse_calls() - n0, st["iterations"], st["hv_evals"], st["not_converged"], kkt.max())
    for prec in ("i8x", "f64", "i8w"):
        print(prec, "bit-identical across forms:", np.array_equal(res[(prec, -1.0)], res[(prec, 1e30)]), "max diff to i8x", np.abs(res[(prec, -1.0)] - res[("i8x", -1.0)]).max())
    # (the FP64 objective passes sum with floating-point atomics: is a repeat of the same FP64 solve bit-identical at all?)
    L.gml_test_hv_sparse_ratio(-1.0)
    a, _, _ = p.learn("RISE", 0.05, tol=1e-9, precision="f64", max_working=128, max_iter=200)
    b, _, _ = p.learn("RISE", 0.05, tol=1e-9, precision="f64", max_working=128, max_iter=200)
    print("f64 repeated with the GEMM form: bit-identical", np.array_equal(a, b), "max diff", np.abs(a - b).max())
    L.gml_test_hv_sparse_ratio(0.3)
