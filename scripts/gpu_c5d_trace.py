"""Config 5 at the default regulariser, full size, with the solver's per-iteration trace on stderr (verbose 2); run under
rocprofv3 --kernel-trace --stats for the kernel split of the direction phase.  usage: gpu_c5d_trace.py [key=value ...]"""
import sys, time
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
kw = {}
for a in sys.argv[1:]:
    k, v = a.split('=')
    kw[k] = float(v) if '.' in v or 'e' in v else int(v)
n, K = 512, 1000000
if 'hvs' in kw:  # switch between the GEMM form and the entry-by-entry form of the Hessian-vector products (test hook)
    import ctypes as C
    L = gml._lib.lib()
    L.gml_test_hv_sparse_ratio.restype = C.c_double
    L.gml_test_hv_sparse_ratio.argtypes = [C.c_double]
    L.gml_test_hv_sparse_ratio(float(kw.pop('hvs')))
terms = syn.block_multibody_terms(n, block=16, seed=0)
if kw.pop('perm', 0):  # the same model with its spins renumbered at random
    import numpy as np
    pm = np.random.default_rng(11).permutation(n)
    terms = {tuple(sorted(int(pm[k - 1]) + 1 for k in key)): v for key, v in terms.items()}
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
    t0 = time.time()
    opts = dict(tol=1e-8, precision="i8x", max_iter=150, verbose=2, raise_on_fail=False)
    opts.update(kw)
    out, kkt, st = p.learn("RISE", 0.4, **opts)
    print("learn_s", time.time() - t0, {k: st[k] for k in ("iterations", "passes", "forward_passes", "hessian_passes", "hv_evals", "t_pass", "t_hess", "t_host", "max_kkt", "not_converged")})
