"""Config 5 at the reference's default regulariser, full size: learn() wall-clock against the sub-sampling of the Hessian-vector
products (gml_opts.hv_subsample; 0 = the automatic rule).  usage: gpu_c5d_sweep.py [values ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
vals = [int(v) for v in sys.argv[1:]] or [0, 1, 4, 8]
n, K = 512, 1000000
terms = syn.block_multibody_terms(n, block=16, seed=0)
res = {}
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
    for v in vals:
        t0 = time.time()
        out, kkt, st = p.learn("RISE", 0.4, tol=1e-8, precision="i8x", max_iter=150, hv_subsample=v, raise_on_fail=False)
        dt = time.time() - t0
        res[str(v)] = {"learn_s": dt, **{k: st[k] for k in ("iterations", "passes", "forward_passes", "hessian_passes", "hv_evals", "node_evals", "t_pass", "t_hess", "t_host", "max_kkt", "not_converged")},
                       "nnz_max": int((out != 0).sum(1).max())}
        print(v, res[str(v)], flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/c5d_sweep.json', 'w'), indent=1)
