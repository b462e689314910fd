"""learn() of the 128-node shard of the headline problem (the per-GPU workload of an 8-GPU run), 6 times: for rocprofv3 --kernel-trace --stats"""
import sys, time
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J = syn.block_ising_model(1024, block=16, seed=0)
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
    for _ in range(6):
        t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision='i8x'); print(round((time.perf_counter() - t) * 1e3, 2), "ms", st["iterations"], st["passes"], flush=True); time.sleep(0.05)
