"""Data points for the other BASELINE configs (run on the GPU box): C2, C3-logRISE, C4 per-rank shard, reduced C5."""
import sys, time, json, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
out = {}

def run(name, spins, form, c, node_range=None, order=2, prec='i8x', tol=1e-9, truth=None):
    t0 = time.time()
    with gml.Problem(spins=spins, node_range=node_range, order=order) as p:
        t_pack = time.time() - t0
        t1 = time.time()
        res, kkt, st = p.learn(form, c, tol=tol, precision=prec, raise_on_fail=False)
        t_learn = time.time() - t1
        km = p.bench_pass(form, res, steps=3, warmup=1, precision=prec) if order == 2 else None
    rec = {'pack_s': t_pack, 'learn_s': t_learn, 'iterations': st['iterations'], 'passes': st['passes'],
           'fwd_passes': st['forward_passes'], 'node_evals': st['node_evals'], 'max_kkt': st['max_kkt'],
           'not_converged': st['not_converged'], 't_pass': st['t_pass'], 't_hess': st['t_hess'], 'pass_ms': km}
    if truth is not None:
        rec['max_err_vs_truth'] = float(np.abs(truth(res)).max())
    out[name] = rec
    print(name, json.dumps(rec), flush=True)

which = sys.argv[1:] or ['c2', 'c3log', 'c4shard', 'c5small']
if 'c2' in which:
    spins, J = syn.block_ising(256, 100000, block=16, seed=0)
    run('C2 n=256 K=1e5 RISE i8x', spins, 'RISE', 0.4, truth=lambda r: 0.5 * (r + r.T) - J)
    run('C2 n=256 K=1e5 RISE f64', spins, 'RISE', 0.4, prec='f64', truth=lambda r: 0.5 * (r + r.T) - J)
if 'c3log' in which:
    spins, J = syn.block_ising(1024, 1000000, block=16, seed=0)
    run('C3 n=1024 K=1e6 logRISE(0.8) i8x', spins, 'logRISE', 0.8, truth=lambda r: 0.5 * (r + r.T) - J)
    run('C3 n=1024 K=1e6 RPLE(0.2) i8x', spins, 'RPLE', 0.2, truth=lambda r: 0.5 * (r + r.T) - J)
if 'c4shard' in which:
    spins, J = syn.block_ising(4096, 1000000, block=8, seed=0)
    run('C4 n=4096 K=1e6 RISE, one rank of 8 (nodes 0..511) i8x', spins, 'RISE', 0.4, node_range=(0, 512),
        truth=lambda r: r[:, :512] - J[:512, :512])
if 'c5small' in which:
    spins, terms = syn.block_multibody(48, 200000, block=12, seed=0)
    run('C5-reduced n=48 order 3 K=2e5 multiRISE i8x (Q=1176 columns)', spins, 'RISE', 0.4, order=3)
json.dump(out, open('gpurun_out/configs.json', 'w'), indent=1)
