"""Where should precision `auto` of gml_learn take the FP64 path?  learn() wall-clock at f64 / i8x / i8w over small problem sizes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
def timeit(p, form, c, prec, tol):
    p.learn(form, c, tol=tol, precision=prec, raise_on_fail=False)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); out, kkt, st = p.learn(form, c, tol=tol, precision=prec, raise_on_fail=False); ts.append(time.perf_counter() - t)
    return sorted(ts)[2] * 1e3, st
# the README example: 3 spins, the 8-row histogram of 1e6 samples
m = np.array([[0.0, 0.1, 0.2], [0.1, 0.0, 0.3], [0.2, 0.3, 0.0]])
hist = syn.enumerate_sample(m, 1000000, seed=0)
cases = [("README 3-spin, 8 rows", dict(samples=hist))]
for n, K, blk in [(9, 512, 9), (16, 4096, 8), (32, 8192, 8), (64, 20000, 8), (64, 100000, 8), (128, 20000, 16), (256, 4096, 16)]:
    spins, J = syn.block_ising(n, K, block=blk, seed=1)
    cases.append((f"n={n} K={K}", dict(spins=spins)))
for name, kw in cases:
    with (gml.Problem(kw["samples"]) if "samples" in kw else gml.Problem(spins=kw["spins"])) as p:
        KPn = p.K * p.P * p.n
        for tol in (1e-9, 1e-11):
            row = []
            for prec in ("f64", "i8x", "i8w", "auto"):
                ms, st = timeit(p, "RISE", 0.4, prec, tol)
                row.append(f"{prec} {ms:7.2f} ms it {st['iterations']:2d}{'*' if st['not_converged'] else ' '}")
            print(f"{name:24s} K*P*n = 2^{np.log2(KPn):4.1f} tol {tol:.0e}: " + "   ".join(row), flush=True)
