import sys, os, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
spins, J = syn.block_ising(512, 300000, block=16, seed=1)
prec = sys.argv[1] if len(sys.argv) > 1 else 'i8x'
with gml.Problem(spins=spins) as p:
    out, kkt, st = p.learn('RISE', 1.5, tol=1e-9, precision=prec, raise_on_fail=False, verbose=int(os.environ.get('V', '1')))
    print({k: st[k] for k in ('iterations', 'passes', 'forward_passes', 'max_kkt', 'not_converged')})
    bad = np.argsort(-kkt)[:5]
    print('worst rows', bad, kkt[bad])
    for r in bad[:2]:
        nz = np.nonzero(out[r])[0]
        print('row', r, 'nz', nz, out[r][nz])
