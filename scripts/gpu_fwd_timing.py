"""Per-workgroup phase times of the i8w forward kernel (timing build: scripts/build_variant.sh timing -DABL_TIMING, run with
GML_LIB_OVERRIDE=gpurun_ab/libgml_timing.so).  s_memrealtime ticks are 100 MHz."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gml_amd as gml
from gml_amd import _lib
synthetic = __import__("importlib").import_module("gml_amd.synthetic")
n, K = 1024, 1000000
J = synthetic.block_ising_model(n, block=16, seed=0)
with gml.Problem(model=J, num_samples=K, seed=0) as p:
    km = p.bench_pass_resident("RISE", J, steps=1, warmup=2, precision="i8w")
    print({k: v for k, v in km.items() if k != "step_ms"})
    L = _lib.lib()
    L.gml_debug_read_vq.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    Kp = (K + 1023) // 1024 * 1024
    per_tile = (Kp // 64) * 6 * 32 * 64
    rows = []
    for tile in (0, 13, 31):
        buf = np.zeros(per_tile, dtype=np.uint8)
        rc = L.gml_debug_read_vq(p._h, tile * per_tile, per_tile, buf.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        img = buf.reshape(Kp // 64, 6 * 32 * 64)
        ts = img[::4, :48].copy().view(np.uint64).reshape(-1, 6)  # one workgroup per 256 samples = 4 images
        rows.append(ts[: K // 256])
    ts = np.concatenate(rows).astype(np.int64)
    t0 = ts[:, 0].min()
    d = np.diff(ts[:, :5], axis=1) * 10.0  # ns
    names = ["sweep A (incl. ring start)", "fold", "sweep B", "epilogue"]
    for j, nm in enumerate(names):
        print(f"{nm:28s} mean {d[:, j].mean() / 1e3:7.2f} us  median {np.median(d[:, j]) / 1e3:7.2f}  p10 {np.percentile(d[:, j], 10) / 1e3:7.2f}  p90 {np.percentile(d[:, j], 90) / 1e3:7.2f}")
    tot = (ts[:, 4] - ts[:, 0]) * 10.0
    print(f"{'whole workgroup':28s} mean {tot.mean() / 1e3:7.2f} us  median {np.median(tot) / 1e3:7.2f}")
    print("kernel span of the sampled workgroups: %.3f ms" % ((ts[:, 4].max() - t0) * 1e-5))
