"""Config 5 at the default regulariser: the Hessian of one learned row on its support, formed explicitly (torch, FP64), and what
block-diagonal preconditioners of the solver's kind do to its spectrum.  Diagnostic for the CG phase of that config:
which entries of the support the slow modes live on.  usage: gpu_c5d_spectrum.py [rows=0,5] [K=1000000] [nodes=32]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
kw = dict(a.split('=') for a in sys.argv[1:])
rows = [int(v) for v in kw.get('rows', '0,5').split(',')]
n, K, B = 512, int(kw.get('K', 1000000)), 16
nodes = int(kw.get('nodes', 32))
t0 = time.time()
spins, terms = syn.block_multibody(n, K, block=B, seed=0)
print('sampled', time.time() - t0, flush=True)
with gml.Problem(spins=spins, order=3, node_range=(0, nodes)) as p:
    t1 = time.time()
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-8, precision="i8x", max_iter=150, verbose=1, raise_on_fail=False)
    print('learn_s', time.time() - t1, {k: st[k] for k in ("iterations", "passes", "hessian_passes", "hv_evals", "max_kkt", "not_converged")}, flush=True)
    keys = {u: p.multi_keys(u) for u in rows}
dev = torch.device('cuda')
S = torch.cat([torch.from_numpy(spins).to(dev).to(torch.float16), torch.ones(K, 1, device=dev, dtype=torch.float16)], 1)  # column n = 1
del spins
np.set_printoptions(precision=4, linewidth=220, suppress=False)


def pcg_steps(H, apply, b, targets=(0.1, 1e-2, 1e-4), steps=120):
    x = torch.zeros_like(b); r = b.clone(); z = apply(r); pp = z.clone(); rz = r @ z; r0 = r.norm(); hist = []
    for i in range(steps):
        Hp = H @ pp; a = rz / (pp @ Hp); x += a * pp; r -= a * Hp; z = apply(r); rzn = r @ z; pp = z + (rzn / rz) * pp; rz = rzn
        hist.append(float(r.norm() / r0))
    return [next((i + 1 for i, v in enumerate(hist) if v < t), None) for t in targets], hist


def block_prec(H, order, T):
    """block-diagonal preconditioner over consecutive groups of T entries of `order`; returns (apply, spectrum of M^-1 H, Linv)"""
    m = H.shape[0]
    Linv = torch.zeros_like(H)
    for s in range(0, m, T):
        idx = order[s:s + T]
        L = torch.linalg.cholesky(H[idx][:, idx])
        Linv[idx[:, None], idx[None, :]] = torch.linalg.inv(L)
    A = Linv @ H @ Linv.T
    ev, V = torch.linalg.eigh(A)
    return (lambda r: Linv.T @ (Linv @ r)), ev, V, Linv


for u in rows:
    th = out[u]
    sup = np.nonzero(th)[0]
    ks = [keys[u][c] for c in sup]
    m = len(sup)
    # statistic of key (u, a, b): s_u s_a s_b; the key lists u first
    ia = torch.tensor([k[1] if len(k) > 1 else n for k in ks], device=dev)
    ib = torch.tensor([k[2] if len(k) > 2 else n for k in ks], device=dev)
    ths = torch.from_numpy(th[sup]).to(dev)
    H = torch.zeros(m, m, device=dev, dtype=torch.float64)
    wsum = 0.0; w2 = 0.0; wmax = 0.0
    CH = 50000
    for s in range(0, K, CH):
        Sc = S[s:s + CH]
        F = (Sc[:, u:u + 1] * Sc[:, ia] * Sc[:, ib]).double()
        w = torch.exp(-(F @ ths))
        wsum += float(w.sum()); w2 += float((w * w).sum()); wmax = max(wmax, float(w.max()))
        H += F.T @ (F * w[:, None])
    H /= K
    blk_u = u // B
    typ = []
    for k in ks:
        o = [v // B for v in k[1:]]
        if len(k) == 1: typ.append('field')
        elif len(k) == 2: typ.append('pair-own' if o[0] == blk_u else 'pair-far')
        else:
            own = sum(v == blk_u for v in o)
            typ.append('tri-own2' if own == 2 else 'tri-own1' if own == 1 else 'tri-same' if o[0] == o[1] else 'tri-diff')
    typ = np.array(typ)
    print('\n=== row', u, 'support', m, 'kkt', kkt[u], 'E[w]', wsum / K, 'E[w^2]/E[w]^2', (w2 / K) / (wsum / K) ** 2, 'max w', wmax)
    print('types', {t: int((typ == t).sum()) for t in sorted(set(typ))})
    ev = torch.linalg.eigvalsh(H)
    print('H: eig lo', ev[:6].cpu().numpy(), 'hi', ev[-6:].cpu().numpy(), 'kappa', float(ev[-1] / ev[0]))
    g = torch.Generator(device=dev); g.manual_seed(0)
    b = torch.randn(m, device=dev, dtype=torch.float64, generator=g)
    dg = torch.diag(H)
    print('jacobi: steps to 0.1/1e-2/1e-4', pcg_steps(H, lambda r: r / dg, b)[0])
    col_order = torch.arange(m, device=dev)
    # cells: entries grouped by the unordered pair of 16-spin blocks of their two spins (field / pair entries: block n/B)
    ba = torch.div(ia, B, rounding_mode='floor'); bb = torch.div(ib, B, rounding_mode='floor')
    cell = torch.minimum(ba, bb) * 64 + torch.maximum(ba, bb)
    cell_order = torch.argsort(cell * 200000 + torch.arange(m, device=dev), stable=True)
    for name, order, T in (('col', col_order, 128), ('col', col_order, 512), ('cell', cell_order, 128), ('cell', cell_order, 512)):
        ap, pev, V, Linv = block_prec(H, order, T)
        steps, hist = pcg_steps(H, ap, b)
        print('tiles %-4s T=%3d: kappa %.1f  lo %s hi %s  steps %s' % (name, T, float(pev[-1] / pev[0]), pev[:5].cpu().numpy(), pev[-5:].cpu().numpy(), steps))
        print('     quantiles 1/10/50/90/99 %%:', torch.quantile(pev, torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99], device=dev, dtype=torch.float64)).cpu().numpy())
        print('     residual history', np.array(hist[:24]))
        if name == 'col' and T == 128:
            # a right-hand side like the solver's: what 16 steps leave of a random one
            x = torch.zeros_like(b); r = b.clone(); z = ap(r); pp = z.clone(); rz = r @ z
            for i in range(16):
                Hp = H @ pp; a = rz / (pp @ Hp); x += a * pp; r -= a * Hp; z = ap(r); rzn = r @ z; pp = z + (rzn / rz) * pp; rz = rzn
            print('     restarted on the residual of 16 steps:', pcg_steps(H, ap, r.clone())[0], np.array(pcg_steps(H, ap, r.clone())[1][:20]))
            # where the extreme modes of the preconditioned matrix live
            for which, js in (('lowest', range(3)), ('highest', range(m - 1, m - 4, -1))):
                for j in js:
                    y = Linv.T @ V[:, j]  # eigenvector in the original coordinates
                    y = y / y.norm()
                    top = torch.argsort(y.abs(), descending=True)[:10].cpu().numpy()
                    mass = {t: float((y.cpu().numpy()[typ == t] ** 2).sum()) for t in sorted(set(typ))}
                    print('     %s mode %d ev %.4f: mass by type %s' % (which, j, float(pev[j]), {k: round(v, 3) for k, v in mass.items()}))
                    print('        top entries', [(ks[i][1:], round(float(y[i]), 3), round(float(th[sup[i]]), 4)) for i in top])
    del H
print('total_s', time.time() - t0)
