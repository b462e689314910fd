"""Soak of round 6's entry points: create / use / destroy in a loop -- gml_learn_terms, gml_learn_matrix, gml_learn_warm, device-pointer
operator calls (objective / gradient / Hessian-vector, int8 and FP64), compacted and dense passes, gml_terms_assemble /
gml_matrix_symmetrize with host and device pointers.  Device memory must return to its starting level, results must repeat bit for bit."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import gml_amd as gml  # noqa: E402

_lib = gml._lib
syn = __import__("importlib").import_module("gml_amd.synthetic")
n3, K3 = 24, 20000
terms = syn.block_multibody_terms(n3, block=12, seed=3)
n2, K2 = 192, 40000
J = syn.block_ising_model(n2, block=16, seed=1)
rng = np.random.default_rng(0)
ref = {}


def same(tag, *arrs):
    cur = [np.array(a, copy=True) for a in arrs]
    if tag in ref:
        assert all(np.array_equal(a, b) for a, b in zip(ref[tag], cur)), tag
    else:
        ref[tag] = cur


free0 = None
for it in range(60):
    with gml.Problem(terms=terms, n=n3, num_samples=K3, seed=2, order=3) as p:
        w, _, st = p.learn("RISE", 0.6, tol=1e-9, terms=bool(it % 2))
        same(f"terms{it % 2}", w)
        rows, _, _ = p.learn("RISE", 0.6, tol=1e-9)
        same("rows3", rows)
        assert np.array_equal(_lib.terms_assemble(rows, n3, 3, bool(it % 2)), w)
        d_rows = torch.from_numpy(rows).cuda()
        assert np.array_equal(_lib.terms_assemble(d_rows.data_ptr(), n3, 3, bool(it % 2), ld=rows.shape[1]), w)
        warm, _, _ = p.learn("RISE", 0.6, tol=1e-9, x0=rows)
        same("warm3", warm)
    with gml.Problem(model=J, num_samples=K2, seed=4) as p:
        form = ("RISE", "logRISE")[it % 2]
        prec = ("i8x", "i8w")[(it // 2) % 2]
        S, _, _ = p.learn(form, 0.4, tol=1e-9, precision=prec, matrix=True)
        same(f"sym-{form}-{prec}", S)
        R, _, _ = p.learn(form, 0.4, tol=1e-9, precision=prec)
        assert np.array_equal(_lib.matrix_symmetrize(R), S)
        nodes = np.arange(n2, dtype=np.int64)
        theta = np.where(np.abs(R) > 0.05, R, 0.0)  # sparse rows: the compacted forward pass
        f_h, g_h = p.objgrad(form, nodes, theta, precision=prec)
        d_th, d_f, d_g = torch.from_numpy(theta).cuda(), torch.zeros(n2, dtype=torch.float64, device="cuda"), \
            torch.zeros((n2, n2), dtype=torch.float64, device="cuda")
        p.objgrad_device(form, nodes, d_th.data_ptr(), n2, d_f.data_ptr(), d_g.data_ptr(), precision=prec)
        assert np.array_equal(d_g.cpu().numpy(), g_h)
        same(f"g-{form}-{prec}", g_h)
        vec = rng.normal(size=theta.shape) if "vec" not in ref else ref["vec"][0]
        same("vec", vec)
        hv = p.hessvec(form, nodes, theta, vec, precision=("i8x", "f64")[it % 3 == 0])
        if it % 3:
            same(f"hv-{form}-{prec}", hv)
        d_v, d_hv = torch.from_numpy(vec).cuda(), torch.zeros_like(d_g)
        p.hessvec_device(form, nodes, d_th.data_ptr(), d_v.data_ptr(), n2, d_hv.data_ptr())
    del d_rows, d_th, d_f, d_g, d_v, d_hv
    torch.cuda.empty_cache()
    _lib.trim_cache()
    free = torch.cuda.mem_get_info()[0]
    if it == 5:
        free0 = free
    if it % 10 == 9:
        print(it, "free GB %.3f (reference level %.3f)" % (free / 1e9, free0 / 1e9), flush=True)
assert abs(free - free0) < 64e6, (free, free0)
print("soak OK: 60 rounds, results bit-identical, device memory back at its level")
