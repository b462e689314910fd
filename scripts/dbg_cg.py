import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 36, 40000
spins, terms = syn.block_multibody(n, K, block=12, seed=3)
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with gml.Problem(spins=spins, order=3) as p:
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-9, precision="i8x", max_working=cap, max_iter=40, verbose=2, raise_on_fail=False)
    print(st)
