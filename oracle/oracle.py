"""ctypes binding of oracle/libgml_oracle.so (gml_oracle.c) plus the host-side glue of the
reference that involves no arithmetic (histogram splitting, multi-body symmetrisation).

TEST INFRASTRUCTURE ONLY -- see gml_oracle.c header.  Reference citations are relative to
/root/reference/src/GraphicalModelLearning.jl.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FORMS = {"RISE": 0, "logRISE": 1, "RPLE": 2}


def build(force=False):
    so = os.path.join(_HERE, "libgml_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("gml_oracle.c", "gml_oracle_fast.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgml_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        i64, dbl, p = C.c_int64, C.c_double, C.c_void_p
        L.gml_oracle_lambda.restype = dbl
        L.gml_oracle_lambda.argtypes = [dbl, i64, dbl]
        L.gml_oracle_multi_nparams.restype = i64
        L.gml_oracle_multi_nparams.argtypes = [i64, C.c_int]
        L.gml_oracle_multi_keys.restype = None
        L.gml_oracle_multi_keys.argtypes = [i64, C.c_int, i64, p]
        L.gml_oracle_objgrad_pair.restype = None
        L.gml_oracle_objgrad_pair.argtypes = [C.c_int, i64, i64, p, p, i64, p, p, p]
        L.gml_oracle_objgrad_rise_nodes.restype = None
        L.gml_oracle_objgrad_rise_nodes.argtypes = [i64, i64, p, p, p, i64, p, p, p]
        L.gml_oracle_learn_pair.restype = dbl
        L.gml_oracle_learn_pair.argtypes = [C.c_int, i64, i64, p, p, dbl, C.c_int, dbl, p, p, p]
        L.gml_oracle_objgrad_multi.restype = None
        L.gml_oracle_objgrad_multi.argtypes = [i64, i64, C.c_int, p, p, i64, p, p, p]
        L.gml_oracle_learn_multi.restype = dbl
        L.gml_oracle_learn_multi.argtypes = [i64, i64, C.c_int, p, p, dbl, dbl, p, p]
        # gml_oracle_fast.c: blocked / OpenMP restatements for full-size checks and the CPU baseline
        L.gml_oracle_objgrad_nodes.restype = None
        L.gml_oracle_objgrad_nodes.argtypes = [C.c_int, i64, i64, p, p, p, i64, p, p, p]
        L.gml_oracle_objgrad_multi3_nodes.restype = None
        L.gml_oracle_objgrad_multi3_nodes.argtypes = [i64, i64, p, p, p, i64, p, p, p]
        L.gml_oracle_learn_pair_fast.restype = dbl
        L.gml_oracle_learn_pair_fast.argtypes = [C.c_int, i64, i64, p, p, i64, i64, dbl, dbl, C.c_int, p, p, p]
        L.gml_oracle_learn_nodes_fast.restype = dbl
        L.gml_oracle_learn_nodes_fast.argtypes = [C.c_int, i64, i64, p, p, p, i64, dbl, dbl, C.c_int, p, p, p]
        L.gml_oracle_set_threads.restype = None
        L.gml_oracle_set_threads.argtypes = [C.c_int]
        L.gml_oracle_set_threads(host_cpus()["threads"])
        _LIB = L
    return _LIB


def host_cpus():
    """CPU resources this process may use: visible logical CPUs, the cgroup CPU quota (a container is often limited to
    far fewer CPUs' worth of time than it sees: 16 of 256 on the GPU box), and the OpenMP thread count the oracle
    uses -- twice the quota (measured best on the GPU box: SMT siblings share the quota), capped by the visible CPUs."""
    visible = os.cpu_count() or 1
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = float(visible)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = min(quota, float(q) / float(per))
    except Exception:
        pass
    env = os.environ.get("GML_ORACLE_THREADS")
    threads = int(env) if env else int(min(visible, max(1, round(2 * quota)) if quota < visible else visible))
    return {"visible": visible, "quota": quota, "threads": threads}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def split_histogram(samples):
    """samples: K x (1+n) histogram, column 0 = counts, columns 1.. = +-1 spins
    (format produced by sampling.jl:52-54, consumed at :76-81).  Returns (counts f64, spins i8)."""
    s = np.asarray(samples)
    counts = np.ascontiguousarray(s[:, 0], dtype=np.float64)
    spins = np.ascontiguousarray(s[:, 1:], dtype=np.int8)
    return counts, spins


def lam(c, n, M):
    return lib().gml_oracle_lambda(float(c), int(n), float(M))


def objgrad_pair(samples, form, u, theta):
    counts, spins = split_histogram(samples)
    K, n = spins.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    f = C.c_double()
    g = np.zeros(n)
    lib().gml_oracle_objgrad_pair(FORMS[form], K, n, _ptr(counts), _ptr(spins), int(u), _ptr(theta),
                                  C.byref(f), _ptr(g))
    return f.value, g


def objgrad_rise_nodes(counts, spins, nodes, theta):
    """fast RISE objgrad for the cpu_baseline timing; theta is len(nodes) x n."""
    K, n = spins.shape
    nodes = np.ascontiguousarray(nodes, dtype=np.int64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    f = np.zeros(len(nodes))
    g = np.zeros((len(nodes), n))
    lib().gml_oracle_objgrad_rise_nodes(K, n, _ptr(counts), _ptr(spins), _ptr(nodes), len(nodes),
                                        _ptr(theta), _ptr(f), _ptr(g))
    return f, g


def objgrad_nodes(form, counts, spins, nodes, theta, want_grad=True):
    """blocked batched objective/gradient (gml_oracle_fast.c) of the pairwise formulations for the listed
    nodes; theta is len(nodes) x n (slot u = field); counts may be None (all ones)."""
    spins = np.ascontiguousarray(spins, dtype=np.int8)
    K, n = spins.shape
    nodes = np.ascontiguousarray(nodes, dtype=np.int64)
    theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(len(nodes), n)
    if counts is not None:
        counts = np.ascontiguousarray(counts, dtype=np.float64)
    f = np.zeros(len(nodes))
    g = np.zeros((len(nodes), n)) if want_grad else None
    lib().gml_oracle_objgrad_nodes(FORMS[form], K, n, None if counts is None else _ptr(counts), _ptr(spins), _ptr(nodes),
                                   len(nodes), _ptr(theta), _ptr(f), None if g is None else _ptr(g))
    return f, g


def objgrad_multi3_nodes(counts, spins, nodes, theta):
    """order-3 multiRISE objective/gradient (gml_oracle_fast.c) of the listed nodes; theta is len(nodes) x P in
    the reference's key order."""
    spins = np.ascontiguousarray(spins, dtype=np.int8)
    K, n = spins.shape
    nodes = np.ascontiguousarray(nodes, dtype=np.int64)
    P = lib().gml_oracle_multi_nparams(n, 3)
    theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(len(nodes), P)
    if counts is not None:
        counts = np.ascontiguousarray(counts, dtype=np.float64)
    f = np.zeros(len(nodes))
    g = np.zeros((len(nodes), P))
    lib().gml_oracle_objgrad_multi3_nodes(K, n, None if counts is None else _ptr(counts), _ptr(spins), _ptr(nodes), len(nodes),
                                          _ptr(theta), _ptr(f), _ptr(g))
    return f, g


def learn_pair_fast(counts, spins, form="RISE", c=None, node_range=None, tol=1e-9, max_iter=100):
    """CPU learn() by the batched working-set Newton method of gml_oracle_fast.c (the device solver's algorithm):
    un-symmetrised rows of the node range.  Returns (rows, kkt, stats dict)."""
    defaults = {"RISE": 0.4, "logRISE": 0.8, "RPLE": 0.2}
    if c is None:
        c = defaults[form]
    spins = np.ascontiguousarray(spins, dtype=np.int8)
    K, n = spins.shape
    n0, n1 = node_range if node_range is not None else (0, n)
    if counts is not None:
        counts = np.ascontiguousarray(counts, dtype=np.float64)
    out = np.zeros((n1 - n0, n))
    kkt = np.zeros(n1 - n0)
    st = np.zeros(3)
    lib().gml_oracle_learn_pair_fast(FORMS[form], K, n, None if counts is None else _ptr(counts), _ptr(spins), n0, n1, float(c),
                                     float(tol), int(max_iter), _ptr(out), _ptr(kkt), _ptr(st))
    return out, kkt, {"iterations": int(st[0]), "passes": int(st[1]), "node_evals": int(st[2])}


def learn_nodes_fast(counts, spins, nodes, form="RISE", c=None, tol=1e-9, max_iter=100):
    """learn_pair_fast for a LIST of nodes (row r = node nodes[r]): the full-size parity tests solve a sample of the nodes."""
    defaults = {"RISE": 0.4, "logRISE": 0.8, "RPLE": 0.2}
    if c is None:
        c = defaults[form]
    spins = np.ascontiguousarray(spins, dtype=np.int8)
    K, n = spins.shape
    nodes = np.ascontiguousarray(nodes, dtype=np.int64)
    if counts is not None:
        counts = np.ascontiguousarray(counts, dtype=np.float64)
    out = np.zeros((len(nodes), n))
    kkt = np.zeros(len(nodes))
    st = np.zeros(3)
    lib().gml_oracle_learn_nodes_fast(FORMS[form], K, n, None if counts is None else _ptr(counts), _ptr(spins), _ptr(nodes), len(nodes), float(c),
                                      float(tol), int(max_iter), _ptr(out), _ptr(kkt), _ptr(st))
    return out, kkt, {"iterations": int(st[0]), "passes": int(st[1]), "node_evals": int(st[2])}


def kkt_residual(x, g, lam, u):
    """max |minimum-norm subgradient| of f + lam * sum_{j != u} |x_j| at x given the smooth gradient g"""
    pg = np.where(x > 0, g + lam, np.where(x < 0, g - lam, np.sign(g) * np.maximum(np.abs(g) - lam, 0)))
    pg[u] = g[u]  # the field slot is not penalised (:171)
    return float(np.abs(pg).max())


def learn_pair(samples, form="RISE", c=None, symmetrize=True, tol=1e-12):
    """learn(samples, RISE/logRISE/RPLE(c, symmetrize)) restated (:154-189, :263-298, :301-336).
    Returns (n x n matrix, per-node KKT residuals, per-node Newton iterations)."""
    defaults = {"RISE": 0.4, "logRISE": 0.8, "RPLE": 0.2}  # :35, :49, :56
    if c is None:
        c = defaults[form]
    counts, spins = split_histogram(samples)
    K, n = spins.shape
    out = np.zeros((n, n))
    kkt = np.zeros(n)
    iters = np.zeros(n, dtype=np.int32)
    lib().gml_oracle_learn_pair(FORMS[form], K, n, _ptr(counts), _ptr(spins), float(c), int(bool(symmetrize)),
                                float(tol), _ptr(out), _ptr(kkt), _ptr(iters))
    return out, kkt, iters


def multi_keys(n, order, u):
    """keys of node u in the reference's construction order (:94-104, models.jl:228-246), 0-based."""
    P = lib().gml_oracle_multi_nparams(n, order)
    keys = np.zeros((P, order), dtype=np.int32)
    lib().gml_oracle_multi_keys(n, order, int(u), _ptr(keys))
    return [tuple(int(v) for v in row if v >= 0) for row in keys]


def objgrad_multi(samples, order, u, theta):
    counts, spins = split_histogram(samples)
    K, n = spins.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    f = C.c_double()
    g = np.zeros(len(theta))
    lib().gml_oracle_objgrad_multi(K, n, int(order), _ptr(counts), _ptr(spins), int(u), _ptr(theta),
                                   C.byref(f), _ptr(g))
    return f.value, g


def learn_multi_rows(samples, c=0.4, order=2, tol=1e-12):
    """The per-node solutions of learn(samples, multiRISE(c, ., order)) before the reconstruction (:94-127): (n x P array in the
    key order of multi_keys, per-node KKT residual)."""
    counts, spins = split_histogram(samples)
    K, n = spins.shape
    P = lib().gml_oracle_multi_nparams(n, order)
    out = np.zeros((n, P))
    kkt = np.zeros(n)
    lib().gml_oracle_learn_multi(K, n, int(order), _ptr(counts), _ptr(spins), float(c), float(tol), _ptr(out), _ptr(kkt))
    return out, kkt


def assemble_multi_dict(rows, keys, symmetrize):
    """The tail of learn(samples, ::multiRISE, ...) restated (:129-151): rows[r] holds the solved parameters of one node in the
    order of keys[r] (0-based tuples (u, ascending others)); returns the reference's Dict {1-based key tuple: value}, symmetrised
    (group by sorted key, `mean`; the members in ascending u) or not."""
    rec = {}
    for kr, xr in zip(keys, rows):
        for key, v in zip(kr, xr):
            rec[tuple(int(i) + 1 for i in key)] = float(v)  # :129-132
    if symmetrize:  # :135-149
        groups = {}
        for k, v in rec.items():
            groups.setdefault(tuple(sorted(k)), []).append(v)
        rec = {k: float(np.mean(v)) for k, v in groups.items()}
    return rec


def listing_order(rec):
    """keys of a model in the order the reference lists them in (models.jl:61,72: sort by (length, key))"""
    return sorted(rec, key=lambda k: (len(k), k))


def learn_multi(samples, c=0.4, symmetrize=True, order=2, tol=1e-12):
    """learn(samples, multiRISE(c, symmetrize, order)) restated (:83-152).
    Returns (dict {1-based key tuple: value}, per-node KKT).  Keys are 1-based like the reference's."""
    counts, spins = split_histogram(samples)
    K, n = spins.shape
    P = lib().gml_oracle_multi_nparams(n, order)
    out = np.zeros((n, P))
    kkt = np.zeros(n)
    lib().gml_oracle_learn_multi(K, n, int(order), _ptr(counts), _ptr(spins), float(c), float(tol), _ptr(out), _ptr(kkt))
    rec = {}
    for u in range(n):
        for key, v in zip(multi_keys(n, order, u), out[u]):
            rec[tuple(i + 1 for i in key)] = float(v)  # :129-132, key = (u, ascending others)
    if symmetrize:  # :135-149  group by sorted key, mean
        groups = {}
        for k, v in rec.items():
            groups.setdefault(tuple(sorted(k)), []).append(v)
        rec = {k: float(np.mean(v)) for k, v in groups.items()}
    return rec, kkt
