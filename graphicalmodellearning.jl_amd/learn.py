"""learn(samples, formulation, method): host-side mirror of the reference's front door
(GraphicalModelLearning.jl:69-73 and the five `learn` methods :83,:154,:210,:263,:301).

All arithmetic over the sample configurations runs in libgml_hip (HIP kernels on MI355X); this
module only marshals arguments, gathers row blocks across ranks and applies the reference's
result assembly (:181-188, :129-151)."""
import warnings

import numpy as np

from . import _lib
from .factor_graph import FactorGraph, TermArray
from .formulations import HIP, NLP, RISE, RISEA, RPLE, GMLFormulation, GMLMethod, logRISE, multiRISE


def _form_name(f):
    return type(f).__name__


def _node_partition(n, world, rank):
    """contiguous node ranges: rank g owns [g*n/G, (g+1)*n/G)  """
    return (rank * n) // world, ((rank + 1) * n) // world


# A learned multi-body model of at most this many terms is handed back with the reference's own container, a dict
# (models.jl:12); above it the FactorGraph keeps the weight array the device assembled (factor_graph.TermArray): config 5's
# 22.5 M terms are 179 MB that way, and ~25 GB / minutes of interpreter time as a dict.
DICT_TERMS_MAX = 1 << 17


def _local_solve_hip(samples, formulation, method, order, node_range, device, terms=None, packed=None, matrix=None):
    """rows of the local node range through libgml_hip: (out, kkt, stats).  terms = True / False (all nodes, multiRISE): the
    model's weight array instead of the rows -- solve and assembly in one library call, the rows never leave the device.
    packed = (sign_bits, counts or None, K): the handle is built from the packed form (gml_problem_create_packed) and `samples`
    is not looked at -- the ranks of a distributed run, which receive the bits from rank 0.  matrix = True / False (all nodes,
    pairwise): the n x n result of gml_learn_matrix, symmetrised on the device (True)."""
    src = {"packed": packed} if packed is not None else {"samples": samples}
    with _lib.Problem(order=order, node_range=node_range, device=device, **src) as prob:
        out, kkt, st = prob.learn(_form_name(formulation), formulation.regularizer, tol=method.tol,
                                  max_iter=method.max_iter, precision=method.precision,
                                  max_working=method.max_working, max_add=method.max_add, verbose=method.verbose,
                                  hess_samples=method.hess_samples, polish=method.polish, terms=terms, matrix=matrix)
    return out, kkt, st


def _symmetrize_hip(R, device):
    """0.5 (R + R') of the gathered rows on the device (gml_matrix_symmetrize; :184-186)"""
    return _lib.matrix_symmetrize(R, device=device)


_PRODUCT_SOLVE = _local_solve_hip  # (what `solve` is unless a test substituted the module attribute)


def _assemble_terms_hip(rows, n, order, symmetrize, device):
    """the gathered rows of all n nodes -> the model's weights in (length, key) order (gml_terms_assemble: one kernel; :129-149)"""
    return _lib.terms_assemble(rows, n, order, symmetrize, device=device)


def _local_solve_multi(samples, formulation, method, order):
    """all nodes on method.devices from this process (gml_multi_*: one handle + one host thread per GPU in the library)"""
    with _lib.MultiProblem(samples, method.devices, order=order) as prob:
        out, kkt, st = prob.learn(_form_name(formulation), formulation.regularizer, tol=method.tol, max_iter=method.max_iter,
                                  precision=method.precision, max_working=method.max_working, max_add=method.max_add,
                                  verbose=method.verbose, hess_samples=method.hess_samples, polish=method.polish)
        st["n_gpus"] = prob.ndev
    return out, kkt, st


def _gather_rows(local, n, P, method):
    """all-gather of the per-rank row blocks (RCCL over xGMI when the backend is nccl)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [(_node_partition(n, world, r)[1] - _node_partition(n, world, r)[0]) for r in range(world)]
    maxr = max(sizes)
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    buf = torch.zeros((maxr, P), dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    allb = torch.empty((world, maxr, P), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allb.view(world * maxr, P), buf)
    allb = allb.cpu().numpy()
    return np.concatenate([allb[r, : sizes[r]] for r in range(world)], axis=0)


def _packed_from_rank0(samples, n_hint=None):
    """The start of a distributed learn ("broadcast of packed spins + weights"): rank 0 reads its sample matrix
    ONCE (gml_pack_histogram: 1 bit per spin, validation in the same sweep), and the sign bits + counts go to the other ranks by
    one broadcast each (RCCL over xGMI under the nccl backend, gloo on CPU).  The other ranks never look at a sample matrix --
    theirs may be None.  At config 4 that is one 32.8 GB read and 0.5 GB per rank on the wire instead of eight concurrent
    packers reading 32.8 GB each.  Returns ((sign_bits, counts or None, K), n, {"pack_s", "bcast_s"})."""
    import time

    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    t0 = time.perf_counter()
    hdr = torch.zeros(4, dtype=torch.int64)  # ok, K, n, counts all one
    bits = counts = None
    err = None
    if rank == 0:
        try:
            bits, counts, _ = _lib.pack_histogram(samples)
            uniform = bool((counts == 1.0).all())
            hdr = torch.tensor([1, len(counts), bits.shape[0], int(uniform)], dtype=torch.int64)
        except Exception as e:  # a bad matrix must fail on EVERY rank, not leave the others waiting in a broadcast
            err = e
    t_pack = time.perf_counter() - t0
    t0 = time.perf_counter()
    hdr = hdr.to(dev)
    dist.broadcast(hdr, 0)
    ok, K, n, uniform = (int(v) for v in hdr.cpu().tolist())
    if not ok:
        raise err if err is not None else _lib.GMLError(_lib.GML_EINVAL, "rank 0 could not pack the sample histogram")
    wpr = int(_lib.lib().gml_packed_words(K))
    tb = torch.from_numpy(bits.view(np.int32)) if rank == 0 else torch.empty((n, wpr), dtype=torch.int32)
    tb = tb.to(dev)
    dist.broadcast(tb, 0)
    if rank != 0:
        bits = tb.cpu().numpy().view(np.uint32)
    if uniform:
        counts = None
    else:
        tc = (torch.from_numpy(counts) if rank == 0 else torch.empty(K, dtype=torch.float64)).to(dev)
        dist.broadcast(tc, 0)
        if rank != 0:
            counts = tc.cpu().numpy()
    t_bcast = time.perf_counter() - t0
    return (bits, counts, K), n, {"pack_s": t_pack, "bcast_s": t_bcast, "bcast_bytes": int(n * wpr * 4 + (0 if uniform else 8 * K))}


def learn(samples, formulation=None, method=None):
    """learn(samples) / learn(samples, formulation) / learn(samples, formulation, method)
    (:69-70: defaults RISE(), NLP()).

    samples: K x (1+n) histogram, column 0 = counts, the rest = +-1 spins (sampling.jl:52-54);
    any real dtype, C or Fortran order, transposed views are copied (:73).
    Returns an n x n ndarray (diagonal = fields) for RISE/logRISE/RPLE/RISEA, a FactorGraph for
    multiRISE."""
    if formulation is None:
        formulation = RISE()
    if method is None:
        method = NLP()
    if not isinstance(formulation, GMLFormulation):
        raise TypeError(f"no method matching learn(::Array, ::{type(formulation).__name__}, ...)")
    if not isinstance(method, GMLMethod):
        raise TypeError(f"no method matching learn(..., ::{type(method).__name__})")
    if isinstance(method, NLP):
        if method.solver is not None:
            # the reference would hand the per-node problems to this optimizer (:111, :164); there is no JuMP here to drive it
            warnings.warn(f"NLP(solver={method.solver!r}): the configured optimizer is not used -- every node-wise problem is solved by the "
                          "device solver of libgml_hip with its default options (pass HIP(...) to set them; to keep an external solver, "
                          "register gml_objgrad_batch as its operator: INTEGRATION.md)", stacklevel=2)
        method = HIP()
    # (argument errors before any collective: every rank raises the same thing at the same point)
    if method.devices is not None and (method.distributed or method.node_range is not None or method.device is not None):
        raise ValueError("HIP: devices (all nodes over several GPUs from this process) excludes distributed, node_range and device")
    if method.distributed and method.node_range is not None:
        raise ValueError("HIP: distributed=True derives the node range from the rank; node_range must not be given")
    if method.precision not in _lib.PRECISIONS:
        raise ValueError(f"HIP: unknown precision {method.precision!r} (use 'auto', 'i8x', 'i8w' or 'f64')")
    order = int(formulation.interaction_order) if isinstance(formulation, multiRISE) else 2
    world, rank = 1, 0
    if method.distributed:
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
    packed, start = None, None
    if method.distributed and world > 1:
        # rank 0 packs its matrix once and broadcasts the bits; the other ranks' `samples` is not looked at (and may be None)
        if rank == 0:
            samples = np.asarray(samples)
            if samples.ndim != 2 or samples.shape[1] < 2:
                samples = None  # (packing reports it, on every rank)
        packed, n, start = _packed_from_rank0(samples if rank == 0 else None)
    else:
        samples = np.asarray(samples)
        if samples.ndim != 2 or samples.shape[1] < 2:
            raise ValueError("samples must be a K x (1+n) histogram matrix")
        n = samples.shape[1] - 1

    node_range = method.node_range or _node_partition(n, world, rank)
    device = method.device
    if device is None:
        device = 0
        if method.distributed:
            import torch
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    # (module attributes, looked up per call: the CPU-only tests of this layer substitute the oracle for both steps)
    solve, assemble, symmetrize = _local_solve_hip, _assemble_terms_hip, _symmetrize_hip
    multi = isinstance(formulation, multiRISE)
    if multi and not (method.distributed and world > 1) and method.devices is None and tuple(node_range) != (0, n):
        raise ValueError("multiRISE assembles a FactorGraph from the rows of ALL nodes (:129-151): solve node shards with "
                         "Problem.learn and hand the gathered rows to _lib.terms_assemble")
    # one process, one GPU, all nodes: solve + assembly in one library call (gml_learn_terms)
    one_call = solve is _PRODUCT_SOLVE and method.devices is None and not (method.distributed and world > 1)
    fused = multi and one_call
    fused_sym = (not multi) and one_call and bool(formulation.symmetrization) and tuple(node_range) == (0, n)  # (gml_learn_matrix)
    try:
        if method.devices is not None:
            out, kkt, st = _local_solve_multi(samples, formulation, method, order)
            node_range = (0, n)
            device = int(list(method.devices)[0])  # (where a multiRISE result is assembled)
        elif fused:
            out, kkt, st = solve(samples, formulation, method, order, node_range, device, terms=bool(formulation.symmetrization))
        elif fused_sym:
            out, kkt, st = solve(samples, formulation, method, order, node_range, device, matrix=True)
        elif packed is not None:
            out, kkt, st = solve(None, formulation, method, order, node_range, device, packed=packed)[:3]
        else:
            out, kkt, st = solve(samples, formulation, method, order, node_range, device)[:3]
    except _lib.GMLConvergenceError as e:  # the reference's @assert (:180): keep what the solver reached
        method.stats.clear()
        method.stats.update(getattr(e, "stats", {}) or {})
        method.stats["kkt"] = getattr(e, "kkt", None)
        raise
    method.stats.clear()
    method.stats.update(st or {})
    method.stats["kkt"] = kkt
    if start is not None:
        method.stats.update(start)  # pack_s (rank 0's one read of the matrix), bcast_s, bcast_bytes

    if method.distributed and world > 1:
        # the ranks may share a GPU with each other and share it with torch's allocator (the gather below): hand the blocks the
        # library keeps for the next handle back to the driver before anybody else needs the memory
        if solve is _PRODUCT_SOLVE:
            _lib.trim_cache()
        P = out.shape[1]
        out = _gather_rows(out, n, P, method)
        node_range = (0, n)

    if multi:
        # :129-151 -- per-key storage, grouping by sorted key and `mean` happen on the device; what comes back is the weight array
        # in the reference's (length, key) listing order, and the keys are never built (factor_graph.TermArray ranks them)
        weights = out if fused else assemble(out, n, order, bool(formulation.symmetrization), device)
        terms = TermArray(n, order, bool(formulation.symmetrization), weights)
        return FactorGraph(order, n, "spin", terms.to_dict() if len(terms) <= DICT_TERMS_MAX else terms)  # :151

    R = np.asarray(out)  # rows node_range; out[u, :] = reconstruction[u, 1:n] (:181)
    if formulation.symmetrization and R.shape[0] == R.shape[1] and not fused_sym:
        R = symmetrize(R, device)  # :184-186, on the device (the host expression walks R' with a stride of n doubles: 0.22 s at n = 4096)
    return R
