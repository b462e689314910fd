"""learn() of the small and headline problems, one line each (A/B of library builds: GML_LIB_OVERRIDE)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
def timed(p, form, c, reps=3, **kw):
    p.learn(form, c, **kw)
    ts = []
    for _ in range(reps):
        t1 = time.perf_counter(); out, kkt, st = p.learn(form, c, **kw); ts.append(time.perf_counter() - t1)
    return round(float(np.median(ts)) * 1e3, 2), st['iterations'], st['passes'], st['forward_passes'], round(st['t_hess'] * 1e3, 2), st['not_converged']
J2 = syn.block_ising_model(256, block=16, seed=0)
with gml.Problem(model=J2, num_samples=100000, seed=0) as p:
    print('C2', timed(p, 'RISE', 0.4, reps=5, tol=1e-9), flush=True)
J = syn.block_ising_model(1024, block=16, seed=0)
for nl in (128, 1024):
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
        for form, c in (('RISE', 0.4), ('logRISE', 0.8)):
            print(nl, form, timed(p, form, c, tol=1e-9, precision='i8x'), flush=True)
