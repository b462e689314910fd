import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 1024, 1000000
spins, J = syn.block_ising(n, K, block=16, seed=0)
form, c = sys.argv[1], float(sys.argv[2])
hs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with gml.Problem(spins=spins) as p:
    res, kkt, st = p.learn(form, c, tol=1e-9, precision='i8x', raise_on_fail=False, verbose=1, hess_samples=hs)
    print({k: st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged','t_pass','t_hess']}, 'nnz/node max', (res != 0).sum(1).max(), 'mean', (res != 0).sum(1).mean(), flush=True)
