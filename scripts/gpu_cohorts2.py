"""Two cohorts of a 128-node shard on ONE GPU, each a handle of its own solved from its own host thread, with the passes on a
low-priority stream and the direction phase on a high-priority one (gml_test_tune 3): does the other cohort's pass hide this cohort's
latency-bound direction phase?  usage: gpu_cohorts2.py [prec]"""
import sys, time, threading, ctypes as C
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
_lib = import_module('gml_amd._lib')
L = _lib.lib()
L.gml_test_tune.restype = C.c_double
L.gml_test_tune.argtypes = [C.c_int, C.c_double]
prec = sys.argv[1] if len(sys.argv) > 1 else 'i8w'
J = syn.block_ising_model(1024, block=16, seed=0)
K = 1000000

def solve_all(probs, reps=5):
    ts = []
    for _ in range(reps):
        out = [None] * len(probs)
        def work(i):
            out[i] = probs[i].learn('RISE', 0.4, tol=1e-9, precision=prec)
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(probs))]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] * 1e3, [o[2]['iterations'] for o in out]

for label, ranges in (("one handle, 128 rows", [(0, 128)]), ("two handles, 64 rows each", [(0, 64), (64, 128)]), ("four handles, 32 rows each", [(0, 32), (32, 64), (64, 96), (96, 128)])):
    probs = [gml.Problem(model=J, num_samples=K, seed=0, node_range=r) for r in ranges]
    for mode in (0, 2, 1):
        L.gml_test_tune(3, float(mode))
        solve_all(probs, 2)
        ms, its = solve_all(probs)
        print(f"{label:28s} streams: {['one', 'high / low priority', 'two, equal priority'][mode]:22s} {ms:7.2f} ms  iterations {its}", flush=True)
    for p in probs: p.close()
L.gml_test_tune(3, 0.0)
