import sys, time
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J2 = syn.block_ising_model(256, block=16, seed=0)
with gml.Problem(model=J2, num_samples=100000, seed=0) as p:
    for _ in range(6):
        t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9); print(round((time.perf_counter() - t) * 1e3, 2), 'ms', st['iterations'], st['passes'], flush=True)
        time.sleep(0.05)
