"""Round 6, VERDICT item 6: config 5 at the reference's default regulariser (ISODUS() = multiRISE(0.4, true, 3), n = 512, K = 1e6,
i8x, tol 1e-8) -- one time-boxed attempt at the two ideas the review named, interleaved with the baseline on one box:
  (b)  lambda-continuation: c = 1.2 -> 0.7 -> 0.4 (and shorter ladders), each stage started from the previous solution
       (gml_learn_warm), the earlier stages to a loose tolerance
  (a') the support admitted on a K/2 strided sub-sample of the configurations (a handle of its own, the same lambda: c / sqrt 2),
       loose tolerance, then the full problem started from that solution: every certifying pass runs at full K
Prints one line per run; the optimum of every variant is compared with the baseline's."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gml_amd as gml  # noqa: E402

syn = __import__("importlib").import_module("gml_amd.synthetic")
n, K = 512, 1000000
terms = syn.block_multibody_terms(n, block=16, seed=0)
KW = dict(precision="i8x", max_iter=150, raise_on_fail=False)


def run(p, c, tol, x0=None):
    t0 = time.time()
    out, kkt, st = p.learn("RISE", c, tol=tol, x0=x0, **KW)
    return out, kkt, st, time.time() - t0


def line(tag, t, st, extra=""):
    print(f"{tag:<46s} {t:7.2f} s  it {st['iterations']:3d}  passes {st['passes']}+{st['forward_passes']}  node-evals {st['node_evals']}  "
          f"H.v evals {st['hv_evals']}  kkt {st['max_kkt']:.2e}  not_conv {st['not_converged']} {extra}", flush=True)


with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
    ref, kkt, st, t = run(p, 0.4, 1e-8)
    line("baseline (first solve of the handle)", t, st)
    base_t = []
    for rep in range(2):
        ref, kkt, st, t = run(p, 0.4, 1e-8)
        base_t.append(t)
        line(f"baseline (warm handle) #{rep}", t, st)
        # (b) ladders
        for ladder, tol_mid in (((1.2, 0.7, 0.4), 1e-5), ((1.2, 0.4), 1e-5), ((0.8, 0.4), 1e-4), ((1.2, 0.8, 0.55, 0.4), 1e-4)):
            x, tot, its, evals, hv = None, 0.0, [], 0, 0
            for c in ladder:
                x, kk, s_, t_ = run(p, c, 1e-8 if c == ladder[-1] else tol_mid, x0=x)
                tot += t_
                its.append(s_["iterations"])
                evals += s_["node_evals"]
                hv += s_["hv_evals"]
            print(f"(b) ladder {ladder} mid-tol {tol_mid:g}: {tot:7.2f} s  iterations {its}  node-evals {evals}  H.v evals {hv}  final kkt "
                  f"{s_['max_kkt']:.2e} not_conv {s_['not_converged']}  max|x - baseline| {np.abs(x - ref).max():.2e}", flush=True)
    spins = p.spins()
    # (a') half of the configurations, strided
    with gml.Problem(spins=spins[::2], order=3) as ph:
        for tol_half in (1e-3, 1e-4):
            xh, kh, sh, th = run(ph, 0.4 / np.sqrt(2.0), tol_half)
            line(f"(a') K/2 handle to {tol_half:g} (first: also allocs)", th, sh)
            xh, kh, sh, th = run(ph, 0.4 / np.sqrt(2.0), tol_half)
            line(f"(a') K/2 handle to {tol_half:g} (warm)", th, sh)
            x, kk, s_, t_ = run(p, 0.4, 1e-8, x0=xh)
            line(f"(a') full problem from the K/2 solution", t_, s_, f" total {th + t_:.2f} s  max|x - baseline| {np.abs(x - ref).max():.2e}")
    ref2, kkt, st, t = run(p, 0.4, 1e-8)
    line("baseline (warm handle) again", t, st, f" max|x - first baseline| {np.abs(ref2 - ref).max():.2e}")
    # sanity of the warm start itself: from the optimum, one certifying pass
    x, kk, s_, t_ = run(p, 0.4, 1e-8, x0=ref)
    line("warm start AT the optimum", t_, s_, f" max|x - baseline| {np.abs(x - ref).max():.2e}")
