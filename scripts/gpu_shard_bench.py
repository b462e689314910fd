"""Per-rank workload of an N-GPU strong-scaling run, timed on ONE GPU: nodes [0, n/N).  The rate it prints is a PROJECTION
(what N ranks would reach if they ran this workload independently and finished together), not a multi-GPU measurement:
those come from `bench.py --gpus N` on a multi-GPU node."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 1024, 1000000
spins, J = syn.block_ising(n, K, block=16, seed=0)
for N in [1, 2, 4, 8]:
    nl = n // N
    with gml.Problem(spins=spins, node_range=(0, nl)) as p:
        th = np.ascontiguousarray(J[:nl])
        p.bench_pass_resident('RISE', th, steps=2, warmup=0, precision='i8x')
        t0 = time.perf_counter(); km = p.bench_pass_resident('RISE', th, steps=10, warmup=0, precision='i8x'); dt = (time.perf_counter() - t0) / 10
        t1 = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision='i8x'); tl = time.perf_counter() - t1
    print(f"N={N}: nodes/rank {nl}: pass wall {dt*1e3:.3f} ms (fwd {km['fwd_ms']:.3f} bwd {km['bwd_ms']:.3f}) -> PROJECTED {n/dt:.0f} node-evals/s for {N} independent ranks (not measured on {N} GPUs); learn {tl:.3f}s", flush=True)
