"""learn() of every 128-node shard of the headline problem, one after the other on one GPU: what each rank of an 8-GPU run solves
(a projection of the driver's SCALE line: learn_wall_s at N = 8 is the slowest shard + the gather).  usage: gpu_shard_all.py [prec] [parts]"""
import sys, time, json
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
prec = sys.argv[1] if len(sys.argv) > 1 else 'i8w'
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, K = 1024, 1000000
J = syn.block_ising_model(n, block=16, seed=0)
res = []
for r in range(parts):
    n0, n1 = r * n // parts, (r + 1) * n // parts
    with gml.Problem(model=J, num_samples=K, seed=0, node_range=(n0, n1)) as p:
        p.learn('RISE', 0.4, tol=1e-9, precision=prec)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec); ts.append(time.perf_counter() - t)
        pm = p.bench_pass_resident('RISE', J[n0:n1], steps=20, warmup=3, precision=prec)
    res.append({"rank": r, "nodes": [n0, n1], "learn_ms": round(sorted(ts)[1] * 1e3, 2), "iterations": st["iterations"], "passes": st["passes"], "fwd": st["forward_passes"],
                "t_pass_ms": round(st["t_pass"] * 1e3, 2), "t_hess_ms": round(st["t_hess"] * 1e3, 2), "pass_ms": round(pm["device_ms_per_pass"], 3)})
    print(res[-1], flush=True)
with gml.Problem(model=J, num_samples=K, seed=0) as p:
    p.learn('RISE', 0.4, tol=1e-9, precision=prec)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec); ts.append(time.perf_counter() - t)
    pm = p.bench_pass_resident('RISE', J, steps=20, warmup=3, precision=prec)
full = {"learn_ms": round(sorted(ts)[1] * 1e3, 2), "iterations": st["iterations"], "passes": st["passes"], "pass_ms": round(pm["device_ms_per_pass"], 3)}
worst = max(q["learn_ms"] for q in res)
print(json.dumps({"precision": prec, "parts": parts, "full_problem": full, "slowest_shard_learn_ms": worst, "projected_learn_scaling": round(full["learn_ms"] / worst, 2),
                  "slowest_shard_pass_ms": max(q["pass_ms"] for q in res), "projected_pass_scaling": round(full["pass_ms"] / max(q["pass_ms"] for q in res), 2)}))
