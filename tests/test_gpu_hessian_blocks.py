"""The working-set Hessian blocks of the int8-limb path (csrc/gml_i8_hess.hip) against a plain numpy statement of the same sums.

The solver's Newton steps are built from H_W = sum_k h_k stat_W stat_W^T over a sub-sample of the configurations (every kstride-th
block of 512), with h_k the curvature weights of the row's last objective pass (RISE / logRISE: w_k exp(-E_k), the second derivative
of :196 / :279; RPLE: 4 w_k sig (1 - sig), of :317).  The kernel carries the weights in 15 bits below the row's largest, dithered;
the blocks are held to 2e-4 of their largest entry here (2e-3 for a row whose weights spread over e-folds, where a few configurations
carry a quarter of the sum and the dither of the others shows: measured 6e-4 .. 9e-4) -- far inside the sampling error of the
sub-sampled sums, which the solver corrects with secant pairs.  (Through a test hook: the blocks never leave the device in a solve.)"""
import ctypes as C

import numpy as np
import pytest

import gml_amd as gml

pytestmark = pytest.mark.gpu
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
_lib = __import__("importlib").import_module("gml_amd._lib")


def _blocks(p, form, prec, nodes, theta, cols, Kh, kstride):
    L = _lib.lib()
    L.gml_test_hessian_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                          C.c_int64, C.c_int64, C.c_void_p]
    nodes = np.ascontiguousarray(nodes, dtype=np.int64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    m = cols.shape[1]
    mp = (m + 31) // 32 * 32
    H = np.zeros((len(nodes), mp, mp))
    _lib.check(L.gml_test_hessian_blocks(p._h, _lib.FORMULATION_IDS[form], _lib.PRECISIONS[prec], len(nodes), _lib._ptr(nodes), _lib._ptr(theta),
                                         theta.shape[1], _lib._ptr(cols), m, Kh, kstride, _lib._ptr(H)))
    return H


@pytest.mark.parametrize("form,prec,m,kstride", [("RISE", "i8x", 70, 1), ("RISE", "i8w", 128, 2), ("logRISE", "i8x", 33, 1),
                                                 ("RPLE", "i8x", 100, 2), ("RPLE", "i8w", 64, 1), ("RISE", "i8x", 200, 2), ("RISE", "i8w", 300, 1)])
def test_working_set_hessian_blocks_match_numpy(form, prec, m, kstride):
    n, K = 320, 20000
    spins, _ = synthetic.block_ising(n, K, block=16, seed=11)
    rng = np.random.default_rng(m)
    nodes = np.array([0, 5, 170, 319, 5])
    theta = rng.normal(scale=0.08, size=(len(nodes), n))
    theta[1] *= 4.0  # a row whose weights spread over a few e-folds
    cols = np.stack([rng.choice(n, size=m, replace=False) for _ in nodes])
    cols[0, 0] = nodes[0]  # the field (slot u: statistic s_u) is in the working set of row 0
    Kp = (K + 1023) // 1024 * 1024
    Kh = (Kp // 512 // kstride) * 512 if kstride > 1 else Kp
    with gml.Problem(spins=spins) as p:
        H = _blocks(p, form, prec, nodes, theta, cols, Kh, kstride)
    S = spins.astype(np.float64)
    keep = np.zeros(K, dtype=bool)
    for cb in range(Kh // 512):
        keep[512 * cb * kstride:512 * cb * kstride + 512] = True
    worst = []
    for r, u in enumerate(nodes):
        stat = S * S[:, [u]]
        stat[:, u] = S[:, u]
        E = stat @ theta[r]
        if form == "RPLE":
            sig = 1.0 / (1.0 + np.exp(2.0 * E))
            h = 4.0 * sig * (1.0 - sig) / K
        else:
            h = np.exp(-E) / K
        W = stat[keep][:, cols[r]]
        ref = (W * h[keep, None]).T @ W
        got = H[r][:m, :m]
        tile_lower = (np.arange(m)[None, :] >> 5) <= (np.arange(m)[:, None] >> 5)  # the kernel fills the lower 32 x 32 tiles
        err = np.abs(got - ref)[tile_lower].max() / np.abs(ref).max()
        worst.append(err)
        assert err <= (2e-3 if r == 1 and form != "RPLE" else 2e-4), (form, prec, m, r, err)
    print(f"{form} {prec} m={m} kstride={kstride}: max |H - ref| / max |ref| per row:", " ".join(f"{e:.1e}" for e in worst))
