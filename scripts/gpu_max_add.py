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
    return round(float(np.median(ts)) * 1e3, 2), st['iterations'], st['passes'], st['forward_passes'], st['not_converged']
J = syn.block_ising_model(1024, block=16, seed=0)
for nl in (128, 1024):
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
        for form, c in (('RISE', 0.4), ('logRISE', 0.8)):
            for ma in (48, 64, 96, 128):
                print(nl, form, 'max_add', ma, timed(p, form, c, tol=1e-9, precision='i8x', max_add=ma), flush=True)
J2 = syn.block_ising_model(256, block=16, seed=0)
with gml.Problem(model=J2, num_samples=100000, seed=0) as p:
    for ma in (48, 64, 96, 128):
        print('C2 max_add', ma, timed(p, 'RISE', 0.4, reps=5, tol=1e-9, max_add=ma), flush=True)
