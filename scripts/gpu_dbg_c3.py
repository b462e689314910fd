import os, subprocess, sys
code = r'''
import sys, time, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J = syn.block_ising_model(1024, block=16, seed=0)
with gml.Problem(model=J, num_samples=1000000, seed=3) as p:
    for form, c in (("logRISE", 0.8), ("RISE", 0.4), ("RPLE", 0.2)):
        out, kkt, st = p.learn(form, c, tol=1e-9, precision="i8x", raise_on_fail=False)
        print(form, 'it', st['iterations'], 'notconv', st['not_converged'], 'kkt', st['max_kkt'], 'bad rows', np.nonzero(kkt > 1e-9)[0][:10], flush=True)
'''
for a in sys.argv[1:]:
    tag, path = a.split('=', 1)
    env = dict(os.environ)
    if path:
        env['GML_LIB_OVERRIDE'] = os.path.abspath(path)
    print('====', tag, flush=True)
    subprocess.run([sys.executable, '-c', code], env=env)
