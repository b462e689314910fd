import sys, time
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
spins, J = syn.block_ising(200, 100000, block=10, seed=1)
with gml.Problem(spins=spins) as p:
    for form, c in [('RISE', 0.4), ('logRISE', 0.2), ('RPLE', 0.2)]:
        for prec in ('f64', 'i8x'):
            p.learn(form, c, tol=1e-9, precision=prec)
            ts = []
            for _ in range(3):
                t = time.perf_counter(); out, kkt, st = p.learn(form, c, tol=1e-9, precision=prec); ts.append(time.perf_counter() - t)
            print(form, c, prec, round(min(ts) * 1e3, 2), 'ms', {k: round(st[k] * 1e3, 2) if k.startswith('t_') else st[k] for k in ('iterations', 'passes', 't_pass', 't_hess', 't_host')}, flush=True)
