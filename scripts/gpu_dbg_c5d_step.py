"""Diagnostic: on config 5 at the default regulariser, is the quadratic model of the Newton-CG step (Hessian-vector operator)
consistent with the objective along the step?  usage: gpu_dbg_c5d_step.py [row ...]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
rows = [int(v) for v in sys.argv[1:]] or [129, 0]
n, K = 512, 1000000
terms = syn.block_multibody_terms(n, block=16, seed=0)
with gml.Problem(terms=terms, n=n, num_samples=K, seed=5, order=3) as p:
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-8, precision="i8x", max_iter=22, raise_on_fail=False)
    lam = st["lambda_"] if "lambda_" in st else 0.4 * np.sqrt(np.log(n * n / 0.05) / K)
    print("lambda", lam, "P", out.shape)
    for r in rows:
        x = out[r].copy()
        nodes = np.array([r])
        f0, g = p.objgrad("RISE", nodes, x[None, :], precision="i8x")
        g = g[0]
        pen = np.ones_like(x, dtype=bool)  # (every statistic of multiRISE is penalised)
        pg = np.where(x != 0, g + lam * np.sign(x), np.where(np.abs(g) > lam, g - lam * np.sign(g), 0.0))
        W = (x != 0) | (pg != 0)
        print(f"row {r}: f {f0[0]:.9f} |W| {W.sum()} nsupp {(x != 0).sum()} kkt {np.abs(pg).max():.3e}")
        def Hv(v):
            vv = np.zeros_like(x); vv[W] = v
            return p.hessvec("RISE", nodes, x[None, :], vv[None, :])[0][W]
        b = -pg[W]
        d = np.zeros_like(b); rr = b.copy(); pp = rr.copy(); rs = rr @ rr; rs0 = rs
        for it in range(40):
            Hp = Hv(pp); a = rs / (pp @ Hp); d += a * pp; rr -= a * Hp; rs2 = rr @ rr
            if it % 5 == 4: print(f"   cg {it}: |r|/|r0| {np.sqrt(rs2 / rs0):.3e}")
            if rs2 < 1e-4 * rs0: break
            pp = rr + rs2 / rs * pp; rs = rs2
        Hd = Hv(d)
        print(f"   true residual |H d + pg|/|pg| {np.linalg.norm(Hd - b) / np.linalg.norm(b):.3e}   g.d {g[W] @ d:.4e}  pg.d {pg[W] @ d:.4e}  d.H.d {d @ Hd:.4e}  |d|_1 {np.abs(d).sum():.3f} max|d| {np.abs(d).max():.3e}")
        dfull = np.zeros_like(x); dfull[W] = d
        # finite-difference check of the operator
        eps = 1e-3
        _, g2 = p.objgrad("RISE", nodes, (x + eps * dfull)[None, :], precision="i8x")
        fd = (g2[0] - g)[W] / eps
        print(f"   operator vs finite differences of the gradient: |fd - Hd|/|Hd| {np.linalg.norm(fd - Hd) / np.linalg.norm(Hd):.3e}   d.fd {d @ fd:.4e}")
        for al in (1.0, 0.5, 0.25, 0.125, 0.0625):
            ft, _ = p.objgrad("RISE", nodes, (x + al * dfull)[None, :], precision="i8x", want_grad=False)
            model = f0[0] + al * (g[W] @ d) + 0.5 * al * al * (d @ Hd)
            print(f"   alpha {al:6.4f}: f {ft[0]:.9f}  model {model:.9f}  (f - f0 {ft[0] - f0[0]:+.3e}, model {model - f0[0]:+.3e})")
