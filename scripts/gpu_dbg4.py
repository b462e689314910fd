import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
spins, J = syn.block_ising(64, 5000, block=16, seed=0)
with gml.Problem(spins=spins) as p:
    ref, _, st0 = p.learn('RISE', 0.4, tol=1e-10, precision='f64', max_working=128)
    print('ref', st0['iterations'], st0['passes'], st0['max_kkt'], 'nnz', (ref != 0).sum(1).max())
    for prec in ['f64', 'i8x']:
        res, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec, max_working=32, raise_on_fail=False, verbose=1)
        print(prec, {k: st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged']}, 'err', np.abs(res-ref).max(), flush=True)
