"""Repeatability soak of the solver's zero-copy staging (round 5): the int8-limb solves are deterministic, so hundreds of learn() calls on
one handle -- a 128-node shard, the whole headline problem, a logRISE and an RPLE solve, two handles interleaved from two threads -- must
return the same bits, iteration counts and pass counts every time."""
import sys, threading, numpy as np
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
J = syn.block_ising_model(1024, block=16, seed=0)
K = 1000000
bad = 0
def soak(p, form, c, prec, reps, label):
    global bad
    ref = None
    for i in range(reps):
        out, kkt, st = p.learn(form, c, tol=1e-9, precision=prec)
        key = (out.tobytes(), st['iterations'], st['passes'], st['forward_passes'], st['node_evals'])
        if ref is None: ref = key
        elif key != ref:
            bad += 1
            print(f"{label}: run {i} differs (iterations {st['iterations']} passes {st['passes']})", flush=True)
    print(f"{label}: {reps} runs, iterations {ref[1]}, passes {ref[2]}+{ref[3]}", flush=True)
with gml.Problem(model=J, num_samples=K, seed=0, node_range=(0, 128)) as p:
    soak(p, 'RISE', 0.4, 'i8w', 300, 'shard i8w')
    soak(p, 'RISE', 0.4, 'i8x', 300, 'shard i8x')
    soak(p, 'logRISE', 0.8, 'i8w', 100, 'shard logRISE i8w')
    soak(p, 'RPLE', 0.2, 'i8x', 100, 'shard RPLE i8x')
with gml.Problem(model=J, num_samples=K, seed=0) as p:
    soak(p, 'RISE', 0.4, 'i8w', 40, 'whole problem i8w')
pa = gml.Problem(model=J, num_samples=K, seed=0, node_range=(0, 64))
pb = gml.Problem(model=J, num_samples=K, seed=0, node_range=(64, 128))
ta = threading.Thread(target=soak, args=(pa, 'RISE', 0.4, 'i8w', 100, 'thread A (rows 0..63)'))
tb = threading.Thread(target=soak, args=(pb, 'RISE', 0.4, 'i8x', 100, 'thread B (rows 64..127)'))
ta.start(); tb.start(); ta.join(); tb.join()
pa.close(); pb.close()
print('differences:', bad)
