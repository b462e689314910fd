"""Cost of the runtime calls a handle creation makes (stream, free-memory query, device allocations), fresh process."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def t(f, *a):
    t0 = time.perf_counter(); rc = f(*a); return (time.perf_counter() - t0) * 1e3, rc
print("hipSetDevice %.2f ms" % t(hip.hipSetDevice, 0)[0])
p0 = C.c_void_p()
print("first hipMalloc(4 KB) %.2f ms" % t(hip.hipMalloc, C.byref(p0), C.c_size_t(4096))[0])
for rep in range(3):
    st = C.c_void_p()
    a = t(hip.hipStreamCreate, C.byref(st))[0]
    fr, tot = C.c_size_t(), C.c_size_t()
    b = t(hip.hipMemGetInfo, C.byref(fr), C.byref(tot))[0]
    ptrs, ms = [], []
    for nbytes in (128 << 20, 128 << 20, 136 << 20, 8 << 20, 1 << 20):
        p = C.c_void_p()
        ms.append(round(t(hip.hipMalloc, C.byref(p), C.c_size_t(nbytes))[0], 2))
        ptrs.append(p)
    fm = [round(t(hip.hipFree, p)[0], 2) for p in ptrs]
    c = t(hip.hipStreamDestroy, st)[0]
    print(f"rep {rep}: hipStreamCreate {a:.2f} ms, hipMemGetInfo {b:.2f} ms, hipMalloc {ms} ms, hipFree {fm} ms, hipStreamDestroy {c:.2f} ms")
h = C.c_void_p()
print("hipHostMalloc(16 MB) %.2f ms" % t(hip.hipHostMalloc, C.byref(h), C.c_size_t(16 << 20), 0)[0])
for rep in range(3):
    for gb in (1, 4, 10.5):
        p = C.c_void_p()
        n = int(gb * 1e9)
        a = t(hip.hipMalloc, C.byref(p), C.c_size_t(n))[0]
        t0 = time.perf_counter(); hip.hipMemset(p, 0, C.c_size_t(n)); hip.hipDeviceSynchronize(); b = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); hip.hipMemset(p, 0, C.c_size_t(n)); hip.hipDeviceSynchronize(); b2 = (time.perf_counter() - t0) * 1e3
        c = t(hip.hipFree, p)[0]
        print(f"rep {rep}: {gb} GB: hipMalloc {a:.1f} ms, first hipMemset {b:.1f} ms, second {b2:.1f} ms, hipFree {c:.1f} ms")
