"""n = 1024 spins on a 32 x 32 periodic lattice (one connected component), 1e6 Glauber chains on the device, RISE."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
Lx = 32
n = Lx * Lx
rng = np.random.default_rng(0)
J = np.zeros((n, n))
for a in range(Lx):
    for b in range(Lx):
        i = a * Lx + b
        for j in (a * Lx + (b + 1) % Lx, ((a + 1) % Lx) * Lx + b):
            J[i, j] = J[j, i] = rng.uniform(0.2, 0.4) * rng.choice([-1, 1])
terms = {(i + 1, j + 1): J[i, j] for i in range(n) for j in range(i + 1, n) if J[i, j] != 0}
t0 = time.time()
with gml.Problem(terms=terms, n=n, num_samples=1000000, seed=1, mcmc_sweeps=int(sys.argv[1]) if len(sys.argv) > 1 else 60) as p:
    print('sample + pack %.2f s' % (time.time() - t0), flush=True)
    for form, c in (('RISE', 0.4), ('logRISE', 0.8), ('RPLE', 0.2)):
        t1 = time.time()
        out, kkt, st = p.learn(form, c, tol=1e-9, precision='i8x', raise_on_fail=False)
        sym = 0.5 * (out + out.T)
        print(form, 'learn %.3f s' % (time.time() - t1), 'it', st['iterations'], 'passes', st['passes'], 'fwd', st['forward_passes'],
              'nc', st['not_converged'], 'nnz max', int((out != 0).sum(1).max()), 'max err %.4f' % np.abs(sym - J).max(), flush=True)
