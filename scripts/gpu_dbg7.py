import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 192, 30000
spins, J = syn.block_ising(n, K, block=16, seed=7)
mw = int(sys.argv[1]); prec = sys.argv[2]
with gml.Problem(spins=spins) as p:
    res, kkt, st = p.learn('RISE', 0.05, tol=1e-9, precision=prec, raise_on_fail=False, verbose=1, max_working=mw, max_iter=400)
    print({k: st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged']}, 'nnz max', (res != 0).sum(1).max(), flush=True)
