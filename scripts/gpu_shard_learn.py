import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 1024, 1000000
J = syn.block_ising_model(n, block=16, seed=0)
for nl, prec in ((128, 'i8x'), (128, 'i8w'), (1024, 'i8x'), (1024, 'i8w')):
    with gml.Problem(model=J, num_samples=K, seed=0, node_range=(0, nl)) as p:
        p.learn('RISE', 0.4, tol=1e-9, precision=prec)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec); ts.append(time.perf_counter() - t1)
    print(nl, prec, [round(t, 4) for t in ts], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items() if k in ('iterations', 'passes', 'forward_passes', 'node_evals', 't_pass', 't_hess', 't_host', 't_total')}, flush=True)
