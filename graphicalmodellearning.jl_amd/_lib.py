"""ctypes binding of libgml_hip.so (C ABI: include/gml.h).

The library is the only compute path: if it is missing, or no HIP device is present, every
call fails loudly -- there is no CPU fallback in the product.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GML_LIB_OVERRIDE") or os.path.join(_HERE, "libgml_hip.so")  # override: A/B builds

GML_OK, GML_EINVAL, GML_ENOTCONV, GML_EHIP, GML_ENOMEM, GML_EUNSUPPORTED = range(6)
GML_ABI_VERSION = 6  # include/gml.h: the revision of the C ABI the mirrors below (Opts, Stats, argument lists) were written against
FORMULATION_IDS = {"RISE": 0, "RISEA": 0, "multiRISE": 0, "logRISE": 1, "RPLE": 2}
DTYPES = {np.dtype(np.int8): 0, np.dtype(np.int32): 1, np.dtype(np.int64): 2, np.dtype(np.float64): 3}
PRECISIONS = {"f64": 0, "i8x": 1, "auto": 2, "i8w": 3}  # auto: i8x, i8w for tight tolerances / small problems / operator calls; i8w: FP64-grade int8 limbs


class GMLError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgml_hip error {code}: {msg}")
        self.code = code


class GMLConvergenceError(AssertionError):
    """Mirrors the reference's `@assert termination_status == LOCALLY_SOLVED`
    (GraphicalModelLearning.jl:127,180,251,289,327)."""


class Opts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int32), ("precision", C.c_int32),
                ("max_working", C.c_int32), ("max_add", C.c_int32), ("verbose", C.c_int32),
                ("hess_samples", C.c_int32), ("polish", C.c_int32), ("max_cg", C.c_int32),
                ("limbs_fwd", C.c_int32), ("hv_limbs_fwd", C.c_int32), ("hv_limbs_bwd", C.c_int32), ("debug_row", C.c_int32),
                ("hv_subsample", C.c_int32), ("coarse", C.c_int32), ("cg_viol_frac", C.c_double), ("cg_eta", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("passes", C.c_int32), ("forward_passes", C.c_int32),
                ("hessian_passes", C.c_int32), ("node_evals", C.c_int64), ("max_kkt", C.c_double),
                ("lambda_", C.c_double), ("t_pack", C.c_double), ("t_pass", C.c_double),
                ("t_hess", C.c_double), ("t_host", C.c_double), ("t_total", C.c_double),
                ("not_converged", C.c_int32), ("polished", C.c_int32), ("hv_evals", C.c_int64), ("t_assemble", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


_lib = None


def lib():
    """Load libgml_hip.so.  torch (if installed) is imported first so that both share one HIP
    runtime (same libamdhip64 SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GMLError(GML_EHIP, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(make -C graphicalmodellearning.jl_amd/csrc)")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64.so first)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    i64, dbl, p, i32 = C.c_int64, C.c_double, C.c_void_p, C.c_int
    _check_abi(L)
    L.gml_last_error.restype = C.c_char_p
    L.gml_default_opts.argtypes = [C.POINTER(Opts)]
    L.gml_default_opts.restype = None
    L.gml_lambda.restype = dbl
    L.gml_lambda.argtypes = [dbl, i64, dbl]
    L.gml_problem_create.argtypes = [p, i32, i64, i64, i64, i32, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_create_device_convert.argtypes = L.gml_problem_create.argtypes
    L.gml_problem_ingest_times.argtypes = [p, p]
    L.gml_packed_words.restype = i64
    L.gml_packed_words.argtypes = [i64]
    L.gml_pack_histogram.argtypes = [p, i32, i64, i64, i64, i32, p, i64, p, C.POINTER(dbl)]
    L.gml_problem_create_packed.argtypes = [p, i64, p, i64, i64, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_get_sign_bits.argtypes = [p, p]
    L.gml_problem_create_spins.argtypes = [p, p, i64, i64, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_create_sampled.argtypes = [p, i64, i64, C.c_uint64, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_create_sampled_terms.argtypes = [p, i32, p, i64, i64, i64, C.c_uint64, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_create_mcmc_terms.argtypes = [p, i32, p, i64, i64, i64, C.c_uint64, i32, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_create_sampled_hist.argtypes = [p, i32, p, i64, i64, i64, C.c_uint64, i32, i32, i64, i64, i32, C.POINTER(p)]
    L.gml_problem_get_counts.argtypes = [p, p]
    L.gml_problem_get_spins.argtypes = [p, p]
    L.gml_problem_destroy.argtypes = [p]
    L.gml_problem_destroy.restype = None
    L.gml_problem_info.argtypes = [p] + [p] * 6
    L.gml_multi_keys.argtypes = [p, i64, p]
    L.gml_objgrad_batch.argtypes = [p, i32, i32, i64, p, p, i64, p, p]
    L.gml_hessvec_batch.argtypes = [p, i32, i64, p, p, p, i64, p]
    L.gml_hessvec_batch_prec.argtypes = [p, i32, i32, i64, p, p, p, i64, p]
    L.gml_multi_create.argtypes = [p, i32, i64, i64, i64, i32, i32, p, i32, C.POINTER(p)]
    L.gml_multi_info.argtypes = [p] + [p] * 6
    L.gml_multi_learn.argtypes = [p, i32, dbl, C.POINTER(Opts), p, p, C.POINTER(Stats), p]
    L.gml_multi_part_stats.argtypes = [p, p]
    L.gml_multi_destroy.argtypes = [p]
    L.gml_multi_destroy.restype = None
    L.gml_learn.argtypes = [p, i32, dbl, C.POINTER(Opts), p, p, C.POINTER(Stats)]
    L.gml_learn_warm.argtypes = [p, i32, dbl, C.POINTER(Opts), p, p, p, C.POINTER(Stats)]
    L.gml_terms_count.restype = i64
    L.gml_terms_count.argtypes = [i64, i32, i32]
    L.gml_terms_assemble.argtypes = [p, i64, i64, i32, i32, i32, p]
    L.gml_terms_keys.argtypes = [i64, i32, i32, i64, i64, p]
    L.gml_terms_rank.restype = i64
    L.gml_terms_rank.argtypes = [i64, i32, i32, p, i32]
    L.gml_learn_terms.argtypes = [p, i32, dbl, i32, C.POINTER(Opts), p, p, C.POINTER(Stats)]
    L.gml_learn_matrix.argtypes = [p, i32, dbl, i32, C.POINTER(Opts), p, p, C.POINTER(Stats)]
    L.gml_matrix_symmetrize.argtypes = [p, i64, i64, i32, p]
    L.gml_bench_pass.argtypes = [p, i32, i32, p, i32, i32, p]
    L.gml_bench_pass_resident.argtypes = [p, i32, i32, p, i32, i32, p, p, p, p]
    _lib = L
    return L


def _check_abi(L):
    """A library of another ABI revision would read / write Opts and Stats at the wrong offsets without any error: refuse it
    (include/gml.h, "ABI identity").  Libraries older than the check itself lack the symbols."""
    try:
        L.gml_sizeof_opts.restype = L.gml_sizeof_stats.restype = C.c_int64
        got = (int(L.gml_abi_version()), int(L.gml_sizeof_opts()), int(L.gml_sizeof_stats()))
    except AttributeError:
        raise GMLError(GML_EUNSUPPORTED, f"{LIB_PATH} predates gml_abi_version(): rebuild it (this binding is ABI {GML_ABI_VERSION})")
    want = (GML_ABI_VERSION, C.sizeof(Opts), C.sizeof(Stats))
    if got != want:
        raise GMLError(GML_EUNSUPPORTED, f"{LIB_PATH} has ABI version / sizeof(gml_opts) / sizeof(gml_stats) = {got}, this binding "
                                         f"was written against {want}: library and binding come from different revisions")


def check(rc, allow=()):
    if rc != GML_OK and rc not in allow:
        raise GMLError(rc, lib().gml_last_error().decode())
    return rc


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _hist_args(samples):
    """(array kept alive, dtype id, K, n, ld, col_major) of a K x (1+n) histogram in the layout it already has"""
    s = np.asarray(samples)
    if s.ndim != 2 or s.shape[1] < 2:
        raise GMLError(GML_EINVAL, "samples must be a K x (1+n) histogram matrix")
    if s.dtype not in DTYPES:
        s = s.astype(np.float64)
    col_major = bool(s.flags.f_contiguous and not s.flags.c_contiguous)
    if not (s.flags.c_contiguous or s.flags.f_contiguous):
        s = np.ascontiguousarray(s)
    K, n = s.shape[0], s.shape[1] - 1
    return s, DTYPES[s.dtype], K, n, (K if col_major else n + 1), int(col_major)


def trim_cache():
    """Return the device memory the library keeps for reuse to the driver (gml_trim_cache); bytes released."""
    L = lib()
    L.gml_trim_cache.restype = C.c_int64
    return int(L.gml_trim_cache())


def set_cache_limit(bytes_per_device):
    """Cap of that cache per device (gml_set_cache_limit): 0 = keep nothing, < 0 = the default (a quarter of the device)."""
    L = lib()
    L.gml_set_cache_limit.restype = None
    L.gml_set_cache_limit.argtypes = [C.c_int64]
    L.gml_set_cache_limit(int(bytes_per_device))


def pack_histogram(samples):
    """Host-only (gml_pack_histogram): histogram matrix -> (sign_bits [n][words] uint32, counts [K] float64, M)."""
    L = lib()
    s, dt, K, n, ld, cm = _hist_args(samples)
    wpr = L.gml_packed_words(K)
    bits = np.empty((n, wpr), dtype=np.uint32)
    counts = np.empty(K, dtype=np.float64)
    M = C.c_double()
    check(L.gml_pack_histogram(_ptr(s), dt, K, n, ld, cm, _ptr(bits), wpr, _ptr(counts), C.byref(M)))
    return bits, counts, M.value


def terms_count(n, order, symmetrize):
    """number of terms of a learned model of `n` spins (gml_terms_count): symmetrised C(n,1)+...+C(n,order), else n P"""
    T = int(lib().gml_terms_count(int(n), int(order), int(bool(symmetrize))))
    if T < 0:
        raise GMLError(GML_EUNSUPPORTED, lib().gml_last_error().decode())
    return T


def terms_assemble(rows, n, order, symmetrize, device=0, ld=None):
    """gml_terms_assemble: the n solved rows -> the model's weights in (length, key) order, on the device.  rows: an (n, P)
    float64 ndarray, or an int device pointer (then ld = its row pitch in doubles)."""
    out = np.empty(terms_count(n, order, symmetrize))
    if isinstance(rows, (int, np.integer)):
        ptr, ld = C.c_void_p(int(rows)), int(ld)
    else:
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if rows.ndim != 2 or rows.shape[0] != n:
            raise GMLError(GML_EINVAL, f"terms_assemble needs the rows of all {n} nodes, got {rows.shape}")
        ptr, ld = _ptr(rows), rows.shape[1]
    check(lib().gml_terms_assemble(ptr, ld, int(n), int(order), int(bool(symmetrize)), int(device), _ptr(out)))
    return out


def matrix_symmetrize(rows, device=0):
    """gml_matrix_symmetrize: 0.5 (R + R') of the gathered n x n rows, on the device (:184-186); same bits as the host expression"""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2 or rows.shape[0] != rows.shape[1]:
        raise GMLError(GML_EINVAL, f"matrix_symmetrize needs the rows of all nodes (a square matrix), got {rows.shape}")
    out = np.empty_like(rows)
    check(lib().gml_matrix_symmetrize(_ptr(rows), rows.shape[1], rows.shape[0], int(device), _ptr(out)))
    return out


def terms_keys(n, order, symmetrize, first=0, count=None):
    """keys of the terms [first, first + count) (gml_terms_keys; host only): int32 [count, order], 0-based spins, -1 = unused"""
    if count is None:
        count = terms_count(n, order, symmetrize) - first
    keys = np.empty((int(count), int(order)), dtype=np.int32)
    check(lib().gml_terms_keys(int(n), int(order), int(bool(symmetrize)), int(first), int(count), _ptr(keys)))
    return keys


def terms_rank(n, order, symmetrize, key0):
    """position of the 0-based key among the terms (gml_terms_rank), -1 if the model has no such key"""
    k = np.ascontiguousarray(key0, dtype=np.int32)
    return int(lib().gml_terms_rank(int(n), int(order), int(bool(symmetrize)), _ptr(k), len(k)))


class Problem:
    """RAII wrapper of a gml_problem handle (packed spins + weights resident in HBM)."""

    def __init__(self, samples=None, *, counts=None, spins=None, packed=None, model=None, terms=None, n=None, num_samples=None,
                 seed=0, mcmc_sweeps=None, order=2, node_range=None, device=0, ingest="host", histogram=False):
        """histogram=True (sampled handles, n <= 64): the handle holds the distinct configurations with their counts
        (gml_problem_create_sampled_hist: sorted and run-length encoded on the device), not one row per draw."""
        L = lib()
        h = C.c_void_p()
        if histogram and (samples is not None or spins is not None or packed is not None):
            raise GMLError(GML_EINVAL, "histogram=True applies to handles sampled on the device (model= or terms=)")
        if histogram and model is not None and terms is None:
            # matrix -> terms exactly as gml_problem_create_sampled lists them (row by row, j <= i: the same seed then gives the
            # same draws), with its symmetry check (models.jl:105-134: the matrix of a FactorGraph is symmetric)
            m = np.asarray(model, dtype=np.float64)
            if m.ndim != 2 or m.shape[0] != m.shape[1] or not np.array_equal(m, m.T):
                raise GMLError(GML_EINVAL, "the model matrix is not symmetric")
            terms = {}
            for i in range(m.shape[0]):
                for j in range(i + 1):
                    if m[i, j] != 0.0:
                        terms[(j + 1, i + 1) if j < i else (i + 1,)] = m[i, j]
            n, model = m.shape[0], None
            if not terms:
                terms = {(1,): 0.0}
        if packed is not None:
            # (sign_bits [n][words] uint32, counts [K] or None, K): the packed form (pack_histogram / sign_bits())
            bits, cnt, K = packed
            bits = np.ascontiguousarray(bits, dtype=np.uint32)
            if cnt is not None:
                cnt = np.ascontiguousarray(cnt, dtype=np.float64)
            n = bits.shape[0]
            n0, n1 = node_range if node_range is not None else (0, n)
            check(L.gml_problem_create_packed(_ptr(bits), bits.shape[1], _ptr(cnt), int(K), n, int(order), n0, n1, int(device),
                                              C.byref(h)))
        elif terms is not None:
            # sample on the device from a model of any order given as {1-based key tuple: weight}: sampling.jl:60-88
            if hasattr(terms, "keys_array"):
                # an array-backed learned model (factor_graph.TermArray: millions of terms): its non-zero terms, without a Python loop
                nz = np.flatnonzero(terms.weights != 0.0)
                if n is None:
                    n = terms.varible_count
                stride = terms.order
                wts = np.ascontiguousarray(terms.weights[nz])
                keys = np.empty((len(nz), stride), dtype=np.int32)
                chunk = 1 << 22
                for a in range(0, len(terms), chunk):  # (the key table is generated window by window: it is never held whole)
                    lo, hi = np.searchsorted(nz, (a, min(len(terms), a + chunk)))
                    if hi > lo:
                        keys[lo:hi] = terms.keys_array(a, min(chunk, len(terms) - a))[nz[lo:hi] - a] - 1
                if len(nz) == 0:
                    keys, wts = np.array([[0] + [-1] * (stride - 1)], dtype=np.int32), np.zeros(1)
            else:
                if n is None:
                    n = max(max(k) for k in terms if len(k))
                stride = max(1, max(len(k) for k in terms))
                keys = np.full((len(terms), stride), -1, dtype=np.int32)
                wts = np.zeros(len(terms), dtype=np.float64)
                for t, (k, v) in enumerate(terms.items()):
                    keys[t, :len(k)] = np.asarray(k, dtype=np.int64) - 1
                    wts[t] = v
            n0, n1 = node_range if node_range is not None else (0, int(n))
            if histogram:
                check(L.gml_problem_create_sampled_hist(_ptr(keys), stride, _ptr(wts), len(wts), int(n), int(num_samples), int(seed),
                                                        int(mcmc_sweeps or 0), int(order), n0, n1, int(device), C.byref(h)))
            elif mcmc_sweeps:  # Glauber chains instead of exact enumeration (components above 22 spins)
                check(L.gml_problem_create_mcmc_terms(_ptr(keys), stride, _ptr(wts), len(wts), int(n), int(num_samples),
                                                      int(seed), int(mcmc_sweeps), int(order), n0, n1, int(device), C.byref(h)))
            else:
                check(L.gml_problem_create_sampled_terms(_ptr(keys), stride, _ptr(wts), len(wts), int(n), int(num_samples),
                                                         int(seed), int(order), n0, n1, int(device), C.byref(h)))
        elif model is not None:
            # sample on the device from a pairwise model (n x n, diagonal = fields): sampling.jl:34-57
            m = np.ascontiguousarray(model, dtype=np.float64)
            n = m.shape[0]
            n0, n1 = node_range if node_range is not None else (0, n)
            check(L.gml_problem_create_sampled(_ptr(m), n, int(num_samples), int(seed), int(order), n0, n1, int(device),
                                               C.byref(h)))
        elif samples is not None:
            s, dt, K, n, ld, cm = _hist_args(samples)
            n0, n1 = node_range if node_range is not None else (0, n)
            # ingest: "host" = packed on the host, bits uploaded (default); "device" = raw upload, converted on the device
            create = {"host": L.gml_problem_create, "device": L.gml_problem_create_device_convert}[ingest]
            check(create(_ptr(s), dt, K, n, ld, cm, int(order), n0, n1, int(device), C.byref(h)))
        else:
            spins = np.ascontiguousarray(spins, dtype=np.int8)
            K, n = spins.shape
            if counts is not None:
                counts = np.ascontiguousarray(counts, dtype=np.float64)
            n0, n1 = node_range if node_range is not None else (0, n)
            check(L.gml_problem_create_spins(_ptr(counts), _ptr(spins), K, n, int(order), n0, n1, int(device),
                                             C.byref(h)))
        self._h = h
        nn, KK, PP, a, b = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        M = C.c_double()
        check(L.gml_problem_info(h, C.byref(nn), C.byref(KK), C.byref(M), C.byref(PP), C.byref(a), C.byref(b)))
        self.n, self.K, self.M, self.P, self.node0, self.node1 = nn.value, KK.value, M.value, PP.value, a.value, b.value
        self.order = int(order)

    def close(self):
        if getattr(self, "_h", None):
            lib().gml_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def spins(self):
        """the +-1 configurations held by the handle (K x n int8)"""
        out = np.zeros((self.K, self.n), dtype=np.int8)
        check(lib().gml_problem_get_spins(self._h, _ptr(out)))
        return out

    def counts(self):
        """the counts of the handle's K rows (column 1 of the histogram, sampling.jl:54)"""
        out = np.zeros(self.K)
        check(lib().gml_problem_get_counts(self._h, _ptr(out)))
        return out

    def sign_bits(self):
        """the packed form of the handle's samples: [n][gml_packed_words(K)] uint32, bit set <=> spin -1"""
        out = np.zeros((self.n, lib().gml_packed_words(self.K)), dtype=np.uint32)
        check(lib().gml_problem_get_sign_bits(self._h, _ptr(out)))
        return out

    def ingest_times(self):
        t = np.zeros(6)
        check(lib().gml_problem_ingest_times(self._h, _ptr(t)))
        return {"pack_s": t[0], "upload_s": t[1], "images_s": t[2], "total_s": t[3], "alloc_s": t[4], "weights_s": t[5]}

    def multi_keys(self, u):
        keys = np.zeros((self.P, self.order), dtype=np.int32)
        check(lib().gml_multi_keys(self._h, int(u), _ptr(keys)))
        return [tuple(int(v) for v in row if v >= 0) for row in keys]

    def objgrad(self, formulation, nodes, theta, precision="auto", want_grad=True):
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(len(nodes), -1)
        f = np.zeros(len(nodes))
        g = np.zeros_like(theta) if want_grad else None
        check(lib().gml_objgrad_batch(self._h, FORMULATION_IDS[formulation], PRECISIONS[precision], len(nodes),
                                      _ptr(nodes), _ptr(theta), theta.shape[1], _ptr(f), _ptr(g)))
        return f, g

    def objgrad_device(self, formulation, nodes, theta_ptr, ld, f_ptr, g_ptr=None, precision="auto"):
        """gml_objgrad_batch on rows resident in HBM: theta_ptr / f_ptr / g_ptr are device addresses (ints: tensor.data_ptr()) of
        [len(nodes), ld], [len(nodes)] and [len(nodes), ld] float64 arrays on the handle's GPU; nothing is staged through the host.
        Ordered after the caller's work on the device's null stream; returns when f and g are written."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        check(lib().gml_objgrad_batch(self._h, FORMULATION_IDS[formulation], PRECISIONS[precision], len(nodes), _ptr(nodes),
                                      C.c_void_p(int(theta_ptr)), int(ld), C.c_void_p(int(f_ptr)),
                                      C.c_void_p(int(g_ptr)) if g_ptr else None))

    def hessvec_device(self, formulation, nodes, theta_ptr, vec_ptr, ld, hv_ptr, precision="i8x"):
        """gml_hessvec_batch_prec on rows resident in HBM (device addresses, as objgrad_device)"""
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        check(lib().gml_hessvec_batch_prec(self._h, FORMULATION_IDS[formulation], PRECISIONS[precision], len(nodes), _ptr(nodes),
                                           C.c_void_p(int(theta_ptr)), C.c_void_p(int(vec_ptr)), int(ld), C.c_void_p(int(hv_ptr))))

    def hessvec(self, formulation, nodes, theta, vec, precision="i8x"):
        """Hess f_u(theta) @ vec for the listed nodes (gml_hessvec_batch_prec): "i8x" = int8-limb passes (~1e-8), "f64" = FP64 MFMA."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(len(nodes), -1)
        vec = np.ascontiguousarray(vec, dtype=np.float64).reshape(len(nodes), -1)
        out = np.zeros_like(theta)
        check(lib().gml_hessvec_batch_prec(self._h, FORMULATION_IDS[formulation], PRECISIONS[precision], len(nodes), _ptr(nodes), _ptr(theta),
                                           _ptr(vec), theta.shape[1], _ptr(out)))
        return out

    def learn(self, formulation, c, *, tol=1e-9, max_iter=100, precision="auto", max_working=512, max_add=64,
              verbose=0, hess_samples=0, polish=True, max_cg=0, limbs_fwd=0, hv_limbs_fwd=0, hv_limbs_bwd=0, debug_row=0,
              hv_subsample=0, cg_viol_frac=0.0, cg_eta=0.0, coarse=True, out_ptr=None, raise_on_fail=True, terms=None, x0=None, matrix=None):
        """gml_learn: (rows, kkt, stats).  terms = True / False (handles over all nodes): gml_learn_terms instead -- the solved
        rows stay on the device and the first result is the model's weight array in (length, key) order, symmetrised (True) or
        not (False): the input of a FactorGraph (TermArray).  x0: rows to start from ((node1-node0) x P, the layout of the result;
        gml_learn_warm) -- a regularisation path solves each c from the previous solution.  matrix = True / False (pairwise handles
        over all nodes): gml_learn_matrix -- the n x n result, symmetrised on the device (True) or as it is (False)."""
        L = lib()
        o = Opts()
        L.gml_default_opts(C.byref(o))
        o.tol, o.max_iter, o.precision = float(tol), int(max_iter), PRECISIONS[precision]
        o.max_working, o.max_add, o.verbose = int(max_working), int(max_add), int(verbose)
        o.hess_samples = int(hess_samples)
        o.polish = 0 if polish else -1
        o.max_cg = int(max_cg)
        o.limbs_fwd, o.hv_limbs_fwd, o.hv_limbs_bwd, o.debug_row = int(limbs_fwd), int(hv_limbs_fwd), int(hv_limbs_bwd), int(debug_row)
        o.cg_viol_frac, o.cg_eta, o.hv_subsample = float(cg_viol_frac), float(cg_eta), int(hv_subsample)
        # precision i8w: early iterations in the cheap 30 / 23-bit form of the pass (True), never (False), or until KKT 10^-coarse
        o.coarse = (0 if coarse else -1) if isinstance(coarse, bool) else int(coarse)
        R = self.node1 - self.node0
        out = None
        kkt = np.zeros(R)
        st = Stats()
        if matrix is not None and (terms is not None or x0 is not None or out_ptr is not None):
            raise GMLError(GML_EINVAL, "matrix (gml_learn_matrix) cannot be combined with terms, x0 or out_ptr")
        if terms is not None and x0 is not None:
            raise GMLError(GML_EINVAL, "x0 (gml_learn_warm) and terms (gml_learn_terms) cannot be combined: solve with x0, then terms_assemble")
        if matrix is not None:
            out = np.zeros((R, self.P))
            rc = L.gml_learn_matrix(self._h, FORMULATION_IDS[formulation], float(c), int(bool(matrix)), C.byref(o), _ptr(out), _ptr(kkt),
                                    C.byref(st))
        elif terms is not None:
            out = np.empty(terms_count(self.n, self.order, terms))
            rc = L.gml_learn_terms(self._h, FORMULATION_IDS[formulation], float(c), int(bool(terms)), C.byref(o), _ptr(out), _ptr(kkt),
                                   C.byref(st))
        else:
            if out_ptr is None:
                out = np.zeros((R, self.P))
                out_ptr = _ptr(out)
            if x0 is not None:
                x0 = np.ascontiguousarray(x0, dtype=np.float64)
                if x0.shape != (R, self.P):
                    raise GMLError(GML_EINVAL, f"x0 has shape {x0.shape}, the handle's rows are {(R, self.P)}")
                rc = L.gml_learn_warm(self._h, FORMULATION_IDS[formulation], float(c), C.byref(o), _ptr(x0), out_ptr, _ptr(kkt), C.byref(st))
            else:
                rc = L.gml_learn(self._h, FORMULATION_IDS[formulation], float(c), C.byref(o), out_ptr, _ptr(kkt), C.byref(st))
        if rc == GML_ENOTCONV:
            if raise_on_fail:
                err = GMLConvergenceError(L.gml_last_error().decode())
                err.stats, err.kkt = st.asdict(), kkt  # what the solver reached, for the caller's diagnostics
                raise err
        else:
            check(rc)
        return out, kkt, st.asdict()

    def multi_keys_array(self, u):
        """gml_multi_keys as it comes: int32 [P, order], 0-based, -1 = unused slot"""
        keys = np.zeros((self.P, self.order), dtype=np.int32)
        check(lib().gml_multi_keys(self._h, int(u), _ptr(keys)))
        return keys

    def bench_pass(self, formulation, theta=None, steps=5, warmup=1, precision="f64"):
        ms = np.zeros(3)
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        check(lib().gml_bench_pass(self._h, FORMULATION_IDS[formulation], PRECISIONS[precision], _ptr(th), int(steps),
                                   int(warmup), _ptr(ms)))
        return {"fwd_ms": ms[0], "bwd_ms": ms[1], "pass_ms": ms[2]}

    def bench_pass_resident(self, formulation, theta, steps=5, warmup=1, precision="f64", want_output=False):
        """Theta uploaded once, then warmup + steps passes back to back with no host round trip
        (gml_bench_pass_resident).  Returns the kernel times and, if asked, (f, g) of the last pass."""
        ms = np.zeros(4)
        th = np.ascontiguousarray(theta, dtype=np.float64)
        nloc = self.node1 - self.node0
        f = np.zeros(nloc) if want_output else None
        g = np.zeros((nloc, self.P)) if want_output else None
        step = np.zeros(int(steps))
        check(lib().gml_bench_pass_resident(self._h, FORMULATION_IDS[formulation], PRECISIONS[precision], _ptr(th), int(steps),
                                            int(warmup), _ptr(ms), _ptr(f), _ptr(g), _ptr(step)))
        out = {"fwd_ms": ms[0], "bwd_ms": ms[1], "pass_ms": ms[2], "device_ms_per_pass": ms[3], "step_ms": step}
        return (out, f, g) if want_output else out


class MultiProblem:
    """One problem on several GPUs of this node from one process (gml_multi_*): GPU g owns the nodes [g n/G, (g+1) n/G),
    one host thread per device inside the library; what the Julia wrapper's HIP(devices = ...) binds."""

    def __init__(self, samples, devices, order=2):
        L = lib()
        s, dt, K, n, ld, cm = _hist_args(samples)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(L.gml_multi_create(_ptr(s), dt, K, n, ld, cm, int(order), _ptr(dev), len(dev), C.byref(h)))
        self._h = h
        nn, KK, PP, nd = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        M = C.c_double()
        check(L.gml_multi_info(h, C.byref(nn), C.byref(KK), C.byref(M), C.byref(PP), C.byref(nd), None))
        self.n, self.K, self.M, self.P, self.ndev = nn.value, KK.value, M.value, PP.value, nd.value
        self.devices = [int(v) for v in dev]

    def part_stats(self):
        """statistics of every part (node shard) of the last learn(): list of dicts"""
        arr = (Stats * self.ndev)()
        check(lib().gml_multi_part_stats(self._h, arr))
        return [a.asdict() for a in arr]

    def diag(self):
        """how the RCCL communicators came about and which path the last gather took (gml_multi_diag)"""
        buf = C.create_string_buffer(256)
        check(lib().gml_multi_diag(self._h, buf, 256))
        return buf.value.decode()

    def gather_kind(self):
        buf = C.create_string_buffer(32)
        check(lib().gml_multi_info(self._h, None, None, None, None, None, buf))
        return buf.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            lib().gml_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def learn(self, formulation, c, *, tol=1e-9, max_iter=100, precision="auto", max_working=512, max_add=64, verbose=0,
              hess_samples=0, polish=True, dev_out=None, raise_on_fail=True):
        """dev_out: optional list of device pointers (ints), one per part, each an n x P float64 buffer on that part's GPU;
        the gathered matrix is left in all of them (RCCL all-gather)."""
        L = lib()
        o = Opts()
        L.gml_default_opts(C.byref(o))
        o.tol, o.max_iter, o.precision = float(tol), int(max_iter), PRECISIONS[precision]
        o.max_working, o.max_add, o.verbose, o.hess_samples = int(max_working), int(max_add), int(verbose), int(hess_samples)
        o.polish = 0 if polish else -1
        out = np.zeros((self.n, self.P))
        kkt = np.zeros(self.n)
        st = Stats()
        dptr = None
        if dev_out is not None:
            dptr = (C.c_void_p * self.ndev)(*[C.c_void_p(int(v)) for v in dev_out])
        rc = L.gml_multi_learn(self._h, FORMULATION_IDS[formulation], float(c), C.byref(o), _ptr(out), _ptr(kkt), C.byref(st), dptr)
        if rc == GML_ENOTCONV:
            if raise_on_fail:
                err = GMLConvergenceError(L.gml_last_error().decode())
                err.stats, err.kkt = st.asdict(), kkt
                raise err
        else:
            check(rc)
        return out, kkt, st.asdict()
