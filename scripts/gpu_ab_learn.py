"""learn() of a few configurations under several builds of the library (argv: tag=path ...; empty path = the tree's build)"""
import json, os, subprocess, sys
code = r'''
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J = syn.block_ising_model(1024, block=16, seed=0)
for nl in (128, 1024):
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
        p.learn('RISE', 0.4, tol=1e-9, precision='i8x')
        t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision='i8x'); dt = time.perf_counter() - t
        print('shard', nl, round(dt * 1e3, 1), 'ms it', st['iterations'], 'passes', st['passes'], 't_hess', round(st['t_hess'] * 1e3, 1), flush=True)
spins, terms = syn.block_multibody(36, 40000, block=12, seed=3)
with gml.Problem(spins=spins, order=3) as p:
    out, kkt, st = p.learn("RISE", 0.4, tol=1e-9, precision="i8x", max_working=64, max_iter=100, raise_on_fail=False)
    print('multibody36 it', st['iterations'], 'notconv', st['not_converged'], 'kkt', st['max_kkt'], 'hv', st['hv_evals'], flush=True)
terms = syn.block_multibody_terms(512, block=16, seed=0)
with gml.Problem(terms=terms, n=512, num_samples=1000000, seed=5, order=3, node_range=(0, 64)) as p:
    t = time.perf_counter(); out, kkt, st = p.learn("RISE", 0.4, tol=1e-8, precision="i8x", max_iter=120, raise_on_fail=False); dt = time.perf_counter() - t
    print('c5 probe 64 nodes', round(dt, 1), 's it', st['iterations'], 'notconv', st['not_converged'], 'kkt', st['max_kkt'], 'hv', st['hv_evals'], 'passes', st['passes'], st['forward_passes'], flush=True)
'''
for a in sys.argv[1:]:
    tag, path = a.split('=', 1)
    env = dict(os.environ)
    if path:
        env['GML_LIB_OVERRIDE'] = os.path.abspath(path)
    print('====', tag, flush=True)
    subprocess.run([sys.executable, '-c', code], env=env)
