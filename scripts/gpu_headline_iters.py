"""Per-iteration trace of learn() at the headline config (verbose 1 on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gml_amd as gml
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
kw = {}
for a in sys.argv[1:]:
    k, v = a.split('=')
    kw[k] = v if not v.lstrip('-').replace('.', '').replace('e-', '').isdigit() else (float(v) if ('.' in v or 'e' in v) else int(v))
n, K = 1024, 1000000
J = synthetic.block_ising_model(n, block=16, seed=0)
with gml.Problem(model=J, num_samples=K, seed=0) as p:
    opts = dict(tol=1e-9, precision="i8w", verbose=1)
    opts.update(kw)
    p.learn("RISE", 0.4, **{**opts, "verbose": 0})
    t0 = time.perf_counter()
    out, kkt, st = p.learn("RISE", 0.4, **opts)
    print("learn_s", time.perf_counter() - t0, {k: st[k] for k in ("iterations", "passes", "forward_passes", "node_evals", "t_pass", "t_hess", "t_host", "max_kkt")})
