"""BASELINE config 5 at full size: n=512 spins, order-3 statistics (130,817 parameters per node), 1e6 samples."""
import sys, time, json, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
t0 = time.time()
spins, terms = syn.block_multibody(n, K, block=16, seed=0)
print('generated', spins.shape, 'in %.1fs' % (time.time() - t0), flush=True)
t0 = time.time()
with gml.Problem(spins=spins, order=3) as p:
    print('P =', p.P, 'create %.1fs' % (time.time() - t0), flush=True)
    th = np.zeros((n, p.P))
    t1 = time.time(); f, g = p.objgrad('RISE', np.arange(n), th, precision='i8x'); t_pass0 = time.time() - t1
    t1 = time.time(); f, g = p.objgrad('RISE', np.arange(n), th, precision='i8x'); t_pass = time.time() - t1
    print('objgrad all nodes: first %.2fs, second %.2fs; f[0]=%.6f' % (t_pass0, t_pass, f[0]), flush=True)
    t1 = time.time()
    creg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
    out, kkt, st = p.learn('RISE', creg, tol=1e-8, precision='i8x', raise_on_fail=False, verbose=1, max_iter=int(sys.argv[4]) if len(sys.argv) > 4 else 100)
    t_learn = time.time() - t1
    keys0 = p.multi_keys(0)
rec = {'n': n, 'K': K, 'c': creg, 'nnz_per_node_max': int((out != 0).sum(1).max()), 'P': int(out.shape[1]), 'pass_s': t_pass, 'learn_s': t_learn, **{k: st[k] for k in ['iterations', 'passes', 'forward_passes', 'hessian_passes', 'max_kkt', 'not_converged', 't_pass', 't_hess', 't_host']}}
# accuracy vs the generating model for node 0's keys
err = 0.0
for key, v in zip(keys0, out[0]):
    k1 = tuple(sorted(i + 1 for i in key))
    err = max(err, abs(v - terms.get(k1, 0.0)))
rec['max_err_node0_vs_truth'] = err
print(json.dumps(rec), flush=True)
json.dump(rec, open('gpurun_out/c5_c%g.json' % creg, 'w'))
