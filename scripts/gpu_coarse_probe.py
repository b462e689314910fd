"""precisions i8w and i8x with and without the coarse early passes: wall-clock, iterations, agreement of the solutions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gml_amd as gml
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
for n, K, form, c in ((1024, 1000000, "RISE", 0.4), (1024, 1000000, "logRISE", 0.8), (256, 100000, "RISE", 0.4)):
    J = synthetic.block_ising_model(n, block=16, seed=0)
    with gml.Problem(model=J, num_samples=K, seed=0) as p:
        res = {}
        for tag, kw in (("i8w coarse", dict(precision="i8w", coarse=True)), ("i8w c 1e-5", dict(precision="i8w", coarse=5)), ("i8w c 1e-7", dict(precision="i8w", coarse=7)), ("i8w c 1e-8", dict(precision="i8w", coarse=8)),
                        ("i8w full", dict(precision="i8w", coarse=False)), ("i8x", dict(precision="i8x")), ("i8x full", dict(precision="i8x", coarse=False)),
                        ("f64", dict(precision="f64"))):
            if tag == "f64" and K > 200000: continue
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); out, kkt, st = p.learn(form, c, tol=1e-9, **kw); ts.append(time.perf_counter() - t0)
            res[tag] = out
            print(f"n={n} K={K} {form:8s} {tag:11s}: {min(ts)*1e3:8.2f} ms  it {st['iterations']} passes {st['passes']}+{st['forward_passes']} evals {st['node_evals']} t_pass {st['t_pass']*1e3:.1f} t_hess {st['t_hess']*1e3:.1f} kkt {st['max_kkt']:.2e} nc {st['not_converged']}", flush=True)
        print("   i8w: max |coarse - full|", np.abs(res["i8w coarse"] - res["i8w full"]).max(), " i8x: max |coarse - full|", np.abs(res["i8x"] - res["i8x full"]).max(),
              " max |i8w - i8x|", np.abs(res["i8w coarse"] - res["i8x"]).max())
