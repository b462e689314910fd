"""Convergence sweep of learn() over models, sizes, regularisers and formulations (all must reach KKT <= tol)."""
import sys, time, itertools, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
bad = 0
cases = []
for n, K, blk in [(64, 20000, 8), (200, 100000, 10), (512, 300000, 16), (1024, 200000, 4)]:
    for seed in (1, 2):
        cases.append((n, K, blk, seed))
for n, K, blk, seed in cases:
    spins, J = syn.block_ising(n, K, block=blk, seed=seed)
    with gml.Problem(spins=spins) as p:
        for form, c in [('RISE', 0.4), ('RISE', 0.1), ('RISE', 1.5), ('logRISE', 0.8), ('logRISE', 0.2), ('RPLE', 0.2), ('RPLE', 1.0)]:
            for prec in (['i8x', 'f64'] if n <= 200 else ['i8x']):
                t0 = time.time()
                out, kkt, st = p.learn(form, c, tol=1e-9, precision=prec, raise_on_fail=False)
                flag = '' if st['not_converged'] == 0 else '  <-- NOT CONVERGED'
                bad += st['not_converged'] != 0
                print(f"n={n} K={K} blk={blk} seed={seed} {form}({c}) {prec}: {time.time()-t0:.3f}s it {st['iterations']} passes {st['passes']}+{st['forward_passes']} "
                      f"kkt {st['max_kkt']:.2e} nnz {int((out != 0).sum(1).max())}{flag}", flush=True)
print('failures:', bad)
