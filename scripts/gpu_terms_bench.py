"""The assembly kernels of the multi-body result (csrc/gml_terms.hip) at config-5 size: n = 512, order 3 -- 67.0 M row entries
(536 MB) -> 22.4 M symmetrised terms (179 MB), rows and result resident in HBM.  HIP-event time per call and the achieved HBM rate
on the algorithmic bytes (3 x 8 B read + 8 B written per triple; the generic per-thread kernel for comparison: argv 'generic' forces
it by calling with order 3 data laid out as order 4 is not possible -- instead the unsymmetrised copy kernel is timed as the
bandwidth reference).  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gml_amd as gml  # noqa: E402

_lib = gml._lib
n, order = 512, 3
P = 1 + (n - 1) + (n - 1) * (n - 2) // 2
rows = torch.randn((n, P), dtype=torch.float64, device="cuda")
L = _lib.lib()
for sym in (1, 0):
    T = _lib.terms_count(n, order, sym)
    out = torch.empty(T, dtype=torch.float64, device="cuda")
    for _ in range(3):
        _lib.check(L.gml_terms_assemble(rows.data_ptr(), P, n, order, sym, 0, out.data_ptr()))
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record()
    for _ in range(reps):
        _lib.check(L.gml_terms_assemble(rows.data_ptr(), P, n, order, sym, 0, out.data_ptr()))
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    nbytes = 8.0 * (n * P + T) if sym else 16.0 * T
    print(f"symmetrize={sym}: {T} terms, {ms:.3f} ms per call (host-blocking calls, null stream), algorithmic bytes {nbytes / 1e6:.0f} MB -> "
          f"{nbytes / ms / 1e6:.0f} GB/s = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
# the whole front-door tail from device rows to a host array (what gml_learn_terms adds to the solve)
t0 = time.perf_counter()
w = _lib.terms_assemble(rows.data_ptr(), n, order, True, ld=P)
print(f"device rows -> host weight array: {(time.perf_counter() - t0) * 1e3:.1f} ms ({w.nbytes / 1e6:.0f} MB to pageable host memory)")
# the pairwise counterpart (k_pair_sym): 0.5 (R + R') at config-4 size, in HBM
for npair in (1024, 4096):
    R = torch.randn((npair, npair), dtype=torch.float64, device="cuda")
    S = torch.empty_like(R)
    for _ in range(3):
        _lib.check(L.gml_matrix_symmetrize(R.data_ptr(), npair, npair, 0, S.data_ptr()))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        _lib.check(L.gml_matrix_symmetrize(R.data_ptr(), npair, npair, 0, S.data_ptr()))
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    nb = 3 * 8.0 * npair * npair  # R read twice (once transposed), S written
    Rh = R.cpu().numpy()
    t0 = time.perf_counter()
    Sh = 0.5 * (Rh + Rh.T)
    t_host = time.perf_counter() - t0
    print(f"pairwise symmetrisation n={npair}: {ms:.3f} ms per call on the device ({nb / ms / 1e6:.0f} GB/s on {nb / 1e6:.0f} MB), "
          f"{t_host * 1e3:.1f} ms for the host expression; same bits: {bool(np.array_equal(S.cpu().numpy(), Sh))}", flush=True)
