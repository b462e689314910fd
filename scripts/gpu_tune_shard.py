"""Experiment sweeps of the solver's knobs (gml_test_tune) on the 128-node shard of the headline problem and on the whole problem.
usage: gpu_tune_shard.py "<id>=<v>,<id>=<v>;<id>=<v>..."   (one setting per ';', '' = defaults)  [prec] [nl list]"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
_lib = import_module('gml_amd._lib')
L = _lib.lib()
L.gml_test_tune.restype = C.c_double
L.gml_test_tune.argtypes = [C.c_int, C.c_double]
settings = sys.argv[1].split(';') if len(sys.argv) > 1 else ['']
prec = sys.argv[2] if len(sys.argv) > 2 else 'i8w'
nls = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [128]
kw = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
J = syn.block_ising_model(1024, block=16, seed=0)
for nl in nls:
    with gml.Problem(model=J, num_samples=1000000, seed=0, node_range=(0, nl)) as p:
        p.learn('RISE', 0.4, tol=1e-9, precision=prec)
        ref = None
        for s in settings:
            for i in range(16): L.gml_test_tune(i, 0.0)
            for kv in [x for x in s.split(',') if x]:
                k, v = kv.split('='); L.gml_test_tune(int(k), float(v))
            ts = []
            for _ in range(5):
                t = time.perf_counter(); out, kkt, st = p.learn('RISE', 0.4, tol=1e-9, precision=prec, **kw); ts.append(time.perf_counter() - t)
            if ref is None: ref = out
            print(f"nl={nl} {prec} [{s}] min {min(ts)*1e3:.2f} med {sorted(ts)[2]*1e3:.2f} ms it {st['iterations']} passes {st['passes']}+{st['forward_passes']} evals {st['node_evals']} t_pass {st['t_pass']*1e3:.2f} t_hess {st['t_hess']*1e3:.2f} t_host {st['t_host']*1e3:.2f} maxkkt {st['max_kkt']:.2e} diff {abs(out-ref).max():.1e}", flush=True)
