"""learn() on the 128-node shard of the headline problem: per-iteration trace (verbose) and timings at i8w / i8x.
usage: gpu_shard_trace.py [nl] [prec] [verbose] [reps]"""
import sys, time
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J = syn.block_ising_model(1024, block=16, seed=0)
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prec = sys.argv[2] if len(sys.argv) > 2 else 'i8w'
verbose = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
    p.learn('RISE', 0.4, tol=1e-9, precision=prec)
    out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec, verbose=verbose)
    for _ in range(reps):
        t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec)
        print(nl, prec, round((time.perf_counter() - t) * 1e3, 2), "ms", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items() if k in ('iterations', 'passes', 'forward_passes', 'node_evals', 't_pass', 't_hess', 't_host')}, flush=True)
        time.sleep(0.05)
