"""Verbose solver trace at the headline config: python scripts/gpu_trace.py [form] [c]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
form = sys.argv[1] if len(sys.argv) > 1 else 'RISE'
c = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rng = np.random.default_rng(7)
n, blk = 1024, 16
from importlib import import_module
syn = import_module('gml_amd.synthetic')
spins, J = syn.block_ising(n, 1000000, block=16, seed=0)
with gml.Problem(spins=spins) as p:
    t0 = time.time()
    out, kkt, st = p.learn(form, c, tol=1e-9, precision='i8x', verbose=1)
    print('learn %.3f s' % (time.time() - t0), st)
