"""The 128- and 256-node shards of the headline problem and config 2 against the Hessian sub-sample (gml_opts.hess_samples; 0 = the
automatic budget) and the working-set growth per iteration (max_add)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')

def timed(p, reps=3, **kw):
    p.learn('RISE', 0.4, **kw)
    ts = []
    for _ in range(reps):
        t1 = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, **kw); ts.append(time.perf_counter() - t1)
    return round(float(np.median(ts)) * 1e3, 2), st['iterations'], st['passes'], round(st['t_pass'] * 1e3, 2), round(st['t_hess'] * 1e3, 2), st['not_converged']

J = syn.block_ising_model(1024, block=16, seed=0)
for nl in (128, 256):
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
        for hs in (0, 65536, 131072, 262144, -1):
            for ma in (64, 128):
                print(nl, 'hess_samples', hs, 'max_add', ma, timed(p, tol=1e-9, precision='i8x', hess_samples=hs, max_add=ma), flush=True)
J2 = syn.block_ising_model(256, block=16, seed=0)
with gml.Problem(model=J2, num_samples=100000, seed=0) as p:
    for hs in (0, 65536, -1):
        for ma in (64, 128):
            print('C2 hess_samples', hs, 'max_add', ma, timed(p, reps=5, tol=1e-9, hess_samples=hs, max_add=ma), flush=True)
