"""Config 5 at the default regulariser on the first `nodes` rows only (device-sampled, as gpu_c5d_trace.py): a 4-second stand-in
for experiments on the solver's CG phase.  usage: gpu_c5d_rows.py [nodes=32] [key=value ...]"""
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
nodes = kw.pop('nodes', 32)
terms = syn.block_multibody_terms(n, block=16, seed=0)
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3, node_range=(0, nodes)) as p:
    opts = dict(tol=1e-8, precision="i8x", max_iter=150, verbose=2, raise_on_fail=False)
    opts.update(kw)
    p.learn("RISE", 0.4, **dict(opts, verbose=0, max_iter=3))  # warm
    t0 = time.time()
    out, kkt, st = p.learn("RISE", 0.4, **opts)
    print("learn_s", time.time() - t0, {k: st[k] for k in ("iterations", "passes", "forward_passes", "hessian_passes", "hv_evals", "t_pass", "t_hess", "max_kkt", "not_converged")})
