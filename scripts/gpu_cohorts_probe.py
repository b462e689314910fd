"""Does running the node loop as several concurrent cohorts on ONE GPU (gml_multi with a repeated device: one handle, host thread
and stream per cohort) hide the latency-bound direction phase behind the other cohorts' passes?  Wall-clock of learn() for
1, 2, 3, 4 cohorts at the sizes where the direction phase weighs most."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gml_amd as gml
synthetic = __import__("importlib").import_module("gml_amd.synthetic")

for n, K, blk in ((256, 100000, 16), (128, 1000000, 16), (256, 1000000, 16), (512, 1000000, 16)):
    J = synthetic.block_ising_model(n, block=blk, seed=0)
    with gml.Problem(model=J, num_samples=K, seed=0) as p:
        spins = p.spins()
        for _ in range(2):
            t0 = time.perf_counter(); o1, _, st = p.learn("RISE", 0.4, tol=1e-9, precision="i8x"); t1 = time.perf_counter() - t0
        print(f"n={n} K={K}: single handle {t1*1e3:.2f} ms (passes {st['t_pass']*1e3:.2f}, direction {st['t_hess']*1e3:.2f}, it {st['iterations']})", flush=True)
    hist = np.empty((K, n + 1), dtype=np.int8)
    hist[:, 0] = 1
    hist[:, 1:] = spins
    for nc in (1, 2, 3, 4):
        with gml.MultiProblem(hist, [0] * nc) as mp:
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); om, _, sm = mp.learn("RISE", 0.4, tol=1e-9, precision="i8x"); ts.append(time.perf_counter() - t0)
            ps = mp.part_stats()
            print(f"   {nc} cohorts: {min(ts)*1e3:.2f} ms  (parts: " + ", ".join(f"{q['t_total']*1e3:.1f}" for q in ps) + f")  max|diff| {np.abs(om - o1).max():.1e}", flush=True)
