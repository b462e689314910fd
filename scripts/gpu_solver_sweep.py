"""Sweep of a solver option (here: hess_samples) on three workloads: python scripts/gpu_solver_sweep.py c4|c3|rple"""
import sys, time, json, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
which = sys.argv[1]
if which == 'c4':
    spins, J = syn.block_ising(4096, 1000000, block=8, seed=0); nr = (0, 512); form, c = 'RISE', 0.4
elif which == 'c3':
    spins, J = syn.block_ising(1024, 1000000, block=16, seed=0); nr = None; form, c = 'RISE', 0.4
else:
    spins, J = syn.block_ising(1024, 1000000, block=16, seed=0); nr = None; form, c = 'RPLE', 0.2
with gml.Problem(spins=spins, node_range=nr) as p:
    for ma in [0, 16384, 32768, 65536, 262144]:
        t1 = time.time()
        res, kkt, st = p.learn(form, c, tol=1e-9, precision='i8x', raise_on_fail=False, max_add=64, hess_samples=ma)
        print(which, 'hess_samples', ma, 'learn %.3f s' % (time.time() - t1), 'it', st['iterations'], 'passes', st['passes'], 'fwd', st['forward_passes'],
              't_pass %.3f t_hess %.3f' % (st['t_pass'], st['t_hess']), 'nc', st['not_converged'], 'nnz max', int((res != 0).sum(1).max()), flush=True)
