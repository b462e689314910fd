"""Handle creation time from a Julia-style histogram matrix (Int64 / Float64, column-major) at the headline size."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spins, J = syn.block_ising(n, K, block=16, seed=0)
for dt in (np.int64, np.float64):
    h = np.empty((K, n + 1), dtype=dt, order='F')
    h[:, 0] = 1
    h[:, 1:] = spins
    t0 = time.time()
    with gml.Problem(h) as p:
        t1 = time.time() - t0
        f, g = p.objgrad('RISE', np.arange(4), J[:4], precision='i8x')
    with gml.Problem(spins=spins) as p:
        f2, g2 = p.objgrad('RISE', np.arange(4), J[:4], precision='i8x')
    print(dt.__name__, 'col-major %.1f GB: create %.2f s' % (h.nbytes / 1e9, t1), 'same result:', np.array_equal(f, f2) and np.array_equal(g, g2), flush=True)
    hc = np.ascontiguousarray(h[:200000])
    t0 = time.time()
    with gml.Problem(hc) as p:
        print(dt.__name__, 'row-major 200k rows: create %.2f s' % (time.time() - t0), flush=True)
    del h, hc
