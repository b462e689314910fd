import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J = syn.block_ising_model(1024, block=16, seed=0)
with gml.Problem(model=J, num_samples=1000000, seed=3, node_range=(64, 96)) as p:
    out, kkt, st = p.learn("logRISE", 0.8, tol=1e-9, precision="i8x", raise_on_fail=False, verbose=2, debug_row=4, max_iter=30)
    print('logRISE it', st['iterations'], 'notconv', st['not_converged'], 'kkt', st['max_kkt'], 'bad rows', np.nonzero(kkt > 1e-9)[0][:10], flush=True)
with gml.Problem(model=J, num_samples=1000000, seed=3, node_range=(0, 32)) as p:
    out, kkt, st = p.learn("RPLE", 0.2, tol=1e-9, precision="i8x", raise_on_fail=False, verbose=1, max_iter=25)
    print('RPLE it', st['iterations'], 'notconv', st['not_converged'], 'kkt', st['max_kkt'], 'hv', st['hv_evals'], flush=True)
