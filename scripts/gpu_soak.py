"""Create / learn / destroy in a loop: device memory must return to its starting level (no leaks), results must repeat."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import gml_amd as gml
from importlib import import_module
syn = import_module('gml_amd.synthetic')
spins, J = syn.block_ising(256, 100000, block=16, seed=0)
hist = np.column_stack([np.ones(len(spins), dtype=np.int64), spins.astype(np.int64)])
spins3, _ = syn.block_multibody(36, 40000, block=12, seed=3)  # order 3, Newton blocks capped below the support: matrix-free rows
free0 = torch.cuda.mem_get_info()[0]
ref = None
ref3 = None
for it in range(40):
    kind = it % 4
    if kind == 0: p = gml.Problem(spins=spins)
    elif kind == 1: p = gml.Problem(np.asfortranarray(hist))
    elif kind == 2: p = gml.Problem(model=J, num_samples=100000, seed=1)
    else: p = gml.Problem(spins=spins, node_range=(64, 192))
    if it % 5 == 4:
        with gml.Problem(spins=spins3, order=3) as p3:
            out3, _, st3 = p3.learn('RISE', 0.4, tol=1e-9, precision='i8x', max_working=64, max_iter=100)
            assert st3['hv_evals'] > 0 and st3['not_converged'] == 0
            if ref3 is None: ref3 = out3.copy()
            assert np.array_equal(ref3, out3)
    with p:
        out, kkt, st = p.learn('RISE' if it % 3 else 'logRISE', 0.4, tol=1e-9, precision='i8x' if it % 2 else 'f64')
        if kind == 0 and it % 3 and it % 2:
            if ref is None: ref = out.copy()
            assert np.array_equal(ref, out)
    gml._lib.trim_cache() if hasattr(gml, '_lib') else import_module('gml_amd._lib').trim_cache()  # (released blocks are cached by the library: hand them back before looking)
    free = torch.cuda.mem_get_info()[0]
    if it == 7: free0 = free  # the runtime's one-time allocations (code objects, pools) are in by now
    if it % 8 == 7: print(it, 'free GB %.3f (start %.3f)' % (free / 1e9, free0 / 1e9), flush=True)
assert abs(free - free0) < 64e6, (free, free0)
try:
    gml.Problem(spins=np.zeros((10, 3), dtype=np.int8))
except gml.GMLError as e:
    print('error path ok:', str(e)[:60])
print('free after error path GB %.3f' % (torch.cuda.mem_get_info()[0] / 1e9))
print('soak OK')
