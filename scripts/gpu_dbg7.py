import sys, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
from oracle import oracle as O
spins, terms = syn.block_multibody(36, 20000, block=12, seed=3)
rng = np.random.default_rng(1)
hist = np.column_stack([np.ones(len(spins), dtype=np.int64), spins.astype(np.int64)])
with gml.Problem(spins=spins, order=3) as p:
    theta = rng.normal(scale=0.05, size=(36, p.P))
    f8, g8 = p.objgrad("RISE", np.arange(36), theta, precision="i8x")
    f64, g64 = p.objgrad("RISE", np.arange(36), theta, precision="f64")
for u in (0, 17, 35):
    fo, go = O.objgrad_multi(hist, 3, u, theta[u])
    print(u, fo, f8[u], f64[u], np.abs(g8[u]-go).max(), np.abs(g64[u]-go).max())
