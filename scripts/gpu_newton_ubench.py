"""k_newton_solve on R blocks of m entries through the test hook (run under rocprofv3 --kernel-trace --stats for its time).
usage: gpu_newton_ubench.py R m [m2 ...]"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from gml_amd import _lib
L = _lib.lib()
L.gml_test_newton_solve.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
R = int(sys.argv[1])
for m in [int(v) for v in sys.argv[2:]]:
    rng = np.random.default_rng(m)
    cap = 512
    X = rng.choice([-1.0, 1.0], size=(4 * m + 50, m)); h = rng.random(len(X))
    A = (X * h[:, None]).T @ X / len(h)
    blocks = np.zeros((R, cap, cap)); blocks[:, :m, :m] = A
    pg = np.zeros((R, cap)); pg[:, :m] = rng.normal(size=(R, m))
    ms = np.full(R, m, dtype=np.int32)
    out = np.zeros((R, cap))
    for _ in range(3):
        _lib.check(L.gml_test_newton_solve(R, _lib._ptr(ms), cap, _lib._ptr(blocks), _lib._ptr(pg), 0.0, None, _lib._ptr(out), 0))
    want = np.linalg.solve(A, -pg[0, :m])
    print(m, "max rel err", np.abs(out[0, :m] - want).max() / np.abs(want).max())
    if np.abs(out[0, 500:506]).max() > 0:  # a -DCHOL_TIMING build: phase times of row 0's workgroup, us
        print("   phases us: load %.1f  diag %.1f  panel %.1f  trailing %.1f  back %.1f  out %.1f" % tuple(out[0, 500:506]))
