"""The solver's trace (verbose 2) on the headline problem: n = 1024, K = 1e6, RISE(0.4).  usage: gpu_headline_trace.py [key=value ...]"""
import sys, time
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
kw = {}
for a in sys.argv[1:]:
    k, v = a.split('=')
    kw[k] = v if k in ('form', 'precision') else (float(v) if '.' in v or 'e' in v else int(v))
form = kw.pop('form', 'RISE')
c = kw.pop('c', 0.4)
J = syn.block_ising_model(1024, block=16, seed=0)
with gml.Problem(model=J, num_samples=1000000, seed=0) as p:
    opts = dict(tol=1e-9, precision='i8x', verbose=0)
    p.learn(form, c, **opts)
    opts.update(kw)
    t0 = time.perf_counter()
    out, kkt, st = p.learn(form, c, **opts)
    print("learn_s", time.perf_counter() - t0, {k: st[k] for k in ("iterations", "passes", "forward_passes", "hessian_passes", "t_pass", "t_hess", "t_host", "max_kkt", "not_converged")})
