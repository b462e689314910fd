"""64-node probe of BASELINE config 5 at the reference's default regulariser (order-3 statistics, n spins, K samples, nodes 0..63):
the dense-optimum path of the solver (matrix-free Newton-CG), cheap enough for parameter studies (GML_CG_VIOL_FRAC, GML_CG_ETA,
GML_CG_MAX).  usage: gpu_c5_probe.py n K [max_working] [max_iter] [verbose]"""
import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = int(sys.argv[1]), int(sys.argv[2])
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 512
mi = int(sys.argv[4]) if len(sys.argv) > 4 else 16
terms = syn.block_multibody_terms(n, block=16, seed=0)
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3, node_range=(0, 64)) as p:
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-8, precision="i8x", max_working=cap, max_iter=mi, verbose=int(sys.argv[5]) if len(sys.argv) > 5 else 2, raise_on_fail=False)
    print(st)
