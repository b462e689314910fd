import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
spins, terms = syn.block_multibody(36, 40000, block=12, seed=3)
with gml.Problem(spins=spins, order=3, node_range=(0, 4)) as p:
    for sub in (1, 0):
        out, kkt, st = p.learn("RISE", 0.4, tol=1e-9, precision="i8x", max_working=64, max_iter=int(sys.argv[1]) if len(sys.argv) > 1 else 12, raise_on_fail=False, verbose=2, hv_subsample=sub)
        print('sub', sub, 'it', st['iterations'], 'notconv', st['not_converged'], 'kkt', st['max_kkt'], 'hv', st['hv_evals'], flush=True)
