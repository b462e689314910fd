"""Pass time of a node-sharded rank's workload (nodes [0, n_loc) of the headline problem) on one GPU.
usage: gpu_shard_pass.py [n_loc ...]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = 1024, 1000000
J = syn.block_ising_model(n, block=16, seed=0)
for nl in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]:
    with gml.Problem(model=J, num_samples=K, seed=0, node_range=(0, nl)) as p:
        th = np.ascontiguousarray(J[:nl])
        p.bench_pass_resident('RISE', th, steps=3, warmup=0, precision='i8x')
        t0 = time.perf_counter(); km = p.bench_pass_resident('RISE', th, steps=40, warmup=0, precision='i8x'); dt = (time.perf_counter() - t0) / 40
    print(f"nodes/rank {nl}: pass wall {dt*1e3:.3f} ms (fwd {km['fwd_ms']:.3f} bwd {km['bwd_ms']:.3f})", flush=True)
