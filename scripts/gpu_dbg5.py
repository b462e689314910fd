import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 1024, 300000
spins, J = syn.block_ising(n, K, block=16, seed=0)
with gml.Problem(spins=spins, node_range=(0, 64)) as p:
    for hs in [0]:
        res, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision='i8x', raise_on_fail=False, verbose=2, hess_samples=hs, max_iter=6)
        print('hess_samples', hs, {k: st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged']}, flush=True)
