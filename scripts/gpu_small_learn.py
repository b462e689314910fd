"""learn() wall-clock where the iteration, not the kernels, is what costs: BASELINE config 1 (README 3-spin model), config 2
(n=256, K=1e5) and the 128-node shard of the headline problem (the per-GPU workload of an 8-GPU run).
Writes gpurun_out/small_learn.json."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')


def timed(p, form, c, reps=5, **kw):
    p.learn(form, c, **kw)
    ts = []
    for _ in range(reps):
        t1 = time.perf_counter()
        out, kkt, st = p.learn(form, c, **kw)
        ts.append(time.perf_counter() - t1)
    keep = ('iterations', 'passes', 'forward_passes', 'node_evals', 't_pass', 't_hess', 't_host', 't_total', 'max_kkt', 'not_converged')
    return {"runs_s": ts, "median_s": float(np.median(ts)), "stats": {k: st[k] for k in keep}}


res = {}
a = np.loadtxt('tests/golden/a_samples.csv', delimiter=',')
with gml.Problem(a) as p:
    res["C1_a_samples_RISE_tol1e-11"] = timed(p, 'RISE', 0.4, tol=1e-11)
J2 = syn.block_ising_model(256, block=16, seed=0)
with gml.Problem(model=J2, num_samples=100000, seed=0) as p:
    res["C2_n256_K1e5_RISE"] = timed(p, 'RISE', 0.4, tol=1e-9)
J = syn.block_ising_model(1024, block=16, seed=0)
for nl in (128, 256, 1024):
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
        res[f"headline_shard_{nl}_nodes"] = timed(p, 'RISE', 0.4, reps=3, tol=1e-9, precision='i8x')
for k, v in res.items():
    print(k, round(v["median_s"] * 1e3, 2), "ms", v["stats"], flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/small_learn.json', 'w'), indent=1)
