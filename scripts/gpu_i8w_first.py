"""First check of precision i8w against the oracle and the FP64 path (objective/gradient at the operator boundary), then
the pass time at the headline shape.  Run on the GPU box: python scripts/gpu_i8w_first.py [n K]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gml_amd as gml
from oracle import oracle as O
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_csv

for name in ["a", "c", "mvt"]:
    s = load_csv(f"{name}_samples.csv")
    n = s.shape[1] - 1
    rng = np.random.default_rng(7)
    theta = rng.normal(scale=0.3, size=(n, n))
    theta[0] = 0.0
    with gml.Problem(s) as p:
        for form in ["RISE", "logRISE", "RPLE"]:
            for prec in ["i8w", "i8x", "f64"]:
                f, g = p.objgrad(form, np.arange(n), theta, precision=prec)
                df = dg = 0.0
                for u in range(n):
                    f0, g0 = O.objgrad_pair(s, form, u, theta[u])
                    df = max(df, abs(f[u] - f0) / max(1.0, abs(f0)))
                    dg = max(dg, np.abs(g[u] - g0).max())
                print(f"{name:4s} {form:8s} {prec}: max df {df:.2e}  max dg {dg:.2e}", flush=True)

n, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1000000)
J = synthetic.block_ising_model(n, block=16, seed=0)
with gml.Problem(model=J, num_samples=K, seed=0) as p:
    theta = np.ascontiguousarray(J)
    res = {}
    for prec in ["i8w", "i8x", "f64"]:
        km, f, g = p.bench_pass_resident("RISE", theta, steps=3 if prec == "f64" else 20, warmup=2, precision=prec, want_output=True)
        res[prec] = (f, g)
        print(prec, {k: (round(v, 4) if isinstance(v, float) else None) for k, v in km.items() if k != "step_ms"}, flush=True)
    for prec in ["i8w", "i8x"]:
        print(prec, "vs f64: max |dg|", np.abs(res[prec][1] - res["f64"][1]).max(), " max rel df", np.abs(res[prec][0] / res["f64"][0] - 1).max())
    spins = p.spins()
    nodes = np.arange(0, n, n // 8, dtype=np.int64)
    f0, g0 = O.objgrad_nodes("RISE", None, spins, nodes, J[nodes])
    for prec in ["i8w", "i8x", "f64"]:
        print(prec, "vs oracle: max |dg|", np.abs(res[prec][1][nodes] - g0).max(), " max rel df", np.abs(res[prec][0][nodes] / f0 - 1).max())
    # random dense-ish theta rows (away from the generating model)
    rng = np.random.default_rng(1)
    th2 = J[nodes] + rng.normal(scale=0.02, size=(len(nodes), n))
    f0, g0 = O.objgrad_nodes("RISE", None, spins, nodes, th2)
    for prec in ["i8w", "i8x", "f64"]:
        f, g = p.objgrad("RISE", nodes, th2, precision=prec)
        print(prec, "dense theta vs oracle: max |dg|", np.abs(g - g0).max(), " max rel df", np.abs(f / f0 - 1).max())
    for form in ["logRISE", "RPLE"]:
        f0, g0 = O.objgrad_nodes(form, None, spins, nodes, J[nodes])
        for prec in ["i8w", "i8x"]:
            f, g = p.objgrad(form, nodes, J[nodes], precision=prec)
            print(form, prec, "vs oracle: max |dg|", np.abs(g - g0).max(), " max abs df", np.abs(f - f0).max())
    t0 = time.perf_counter()
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-9, precision="i8w", raise_on_fail=False)
    print("learn i8w", time.perf_counter() - t0, {k: st[k] for k in ("iterations", "passes", "forward_passes", "max_kkt", "not_converged", "t_pass", "t_hess", "polished")})
    t0 = time.perf_counter()
    out2, kkt2, st2 = p.learn("RISE", 0.4, tol=1e-9, precision="i8x", raise_on_fail=False)
    print("learn i8x", time.perf_counter() - t0, {k: st2[k] for k in ("iterations", "passes", "forward_passes", "max_kkt", "not_converged", "t_pass", "t_hess", "polished")})
    print("max |i8w - i8x| solution", np.abs(out - out2).max())
