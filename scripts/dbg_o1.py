import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gml_amd as gml
s = np.loadtxt('tests/golden/c_samples.csv', delimiter=',')
with gml.Problem(s, order=1) as p:
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-11, precision=sys.argv[1], verbose=2, raise_on_fail=False)
    print(out.ravel(), kkt, st['passes'], st['forward_passes'])
counts, spins = s[:, 0], s[:, 1:]
w = counts / counts.sum()
print([0.5 * np.log(w[spins[:, u] > 0].sum() / w[spins[:, u] < 0].sum()) for u in range(4)])
