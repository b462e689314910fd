import sys, time, json, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
which = sys.argv[1]
if which == 'rple':
    spins, J = syn.block_ising(256, 100000, block=16, seed=0)
    for prec in ['f64', 'i8x']:
        with gml.Problem(spins=spins) as p:
            res, kkt, st = p.learn('RPLE', 0.2, tol=1e-9, precision=prec, raise_on_fail=False, verbose=1)
            print(prec, {k: st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged']}, flush=True)
if which == 'c4':
    spins, J = syn.block_ising(1024, 200000, block=8, seed=0)
    for prec in ['f64']:
        with gml.Problem(spins=spins, node_range=(0, 128)) as p:
            res, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec, raise_on_fail=False, verbose=1)
            print(prec, {k: st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged']}, 'nnz/node', (res != 0).sum(1).mean(), flush=True)
