"""Where does the time of gml_problem_create from a host Matrix{Int64} go on a fresh box?  (alloc / weights / copy waits)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from gml_amd import _lib
K, n = 1000000, 1024
rng = np.random.default_rng(0)
h = np.empty((K, n + 1), dtype=np.int64, order='F')
h[:, 0] = 1
for j0 in range(0, n, 64):
    h[:, 1 + j0:1 + j0 + 64] = rng.integers(0, 2, size=(K, 64), dtype=np.int8) * 2 - 1
for rep in range(5):
    t0 = time.perf_counter()
    p = _lib.Problem(h)
    t1 = time.perf_counter()
    print(rep, round(t1 - t0, 4), {k: round(v, 4) for k, v in p.ingest_times().items()}, flush=True)
    if rep == 2:
        out, kkt, st = p.learn('RISE', 0.4, tol=1e-6, raise_on_fail=False)
    t2 = time.perf_counter()
    p.close()
    print('   close', round(time.perf_counter() - t2, 4), flush=True)
