"""learn(samples, formulation, method): host-side mirror of the reference's front door
(GraphicalModelLearning.jl:69-73 and the five `learn` methods :83,:154,:210,:263,:301).

All arithmetic over the sample configurations runs in libgml_hip (HIP kernels on MI355X); this
module only marshals arguments, gathers row blocks across ranks and applies the reference's
result assembly (:181-188, :129-151)."""
import numpy as np

from . import _lib
from .factor_graph import FactorGraph
from .formulations import HIP, NLP, RISE, RISEA, RPLE, GMLFormulation, GMLMethod, logRISE, multiRISE


def _form_name(f):
    return type(f).__name__


def _node_partition(n, world, rank):
    """contiguous node ranges: rank g owns [g*n/G, (g+1)*n/G)  """
    return (rank * n) // world, ((rank + 1) * n) // world


def _local_solve_hip(samples, formulation, method, order, node_range, device):
    """rows of the local node range through libgml_hip: (out, kkt, stats, keys)"""
    with _lib.Problem(samples, order=order, node_range=node_range, device=device) as prob:
        out, kkt, st = prob.learn(_form_name(formulation), formulation.regularizer, tol=method.tol,
                                  max_iter=method.max_iter, precision=method.precision,
                                  max_working=method.max_working, max_add=method.max_add, verbose=method.verbose,
                                  hess_samples=method.hess_samples, polish=method.polish)
        keys = None
        if isinstance(formulation, multiRISE):
            if order == 2:  # the C ABI keeps the pairwise slot layout for order 2 (slot u = field)
                n = prob.n
                keys = [[(u,) if i == u else (u, i) for i in range(n)] for u in range(node_range[0], node_range[1])]
            else:
                keys = [prob.multi_keys(u) for u in range(node_range[0], node_range[1])]
    return out, kkt, st, keys


_local_solve_hip.pairwise_slots = True  # the C ABI keeps the pairwise slot layout for order 2 (slot u = field)


def _local_solve_multi(samples, formulation, method, order):
    """all nodes on method.devices from this process (gml_multi_*: one handle + one host thread per GPU in the library)"""
    with _lib.MultiProblem(samples, method.devices, order=order) as prob:
        out, kkt, st = prob.learn(_form_name(formulation), formulation.regularizer, tol=method.tol, max_iter=method.max_iter,
                                  precision=method.precision, max_working=method.max_working, max_add=method.max_add,
                                  verbose=method.verbose, hess_samples=method.hess_samples, polish=method.polish)
        st["n_gpus"] = prob.ndev
    return out, kkt, st, None


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
        method = HIP()
    samples = np.asarray(samples)
    if samples.ndim != 2 or samples.shape[1] < 2:
        raise ValueError("samples must be a K x (1+n) histogram matrix")
    n = samples.shape[1] - 1
    order = int(formulation.interaction_order) if isinstance(formulation, multiRISE) else 2

    if method.devices is not None and (method.distributed or method.node_range is not None or method.device is not None):
        raise ValueError("HIP: devices (all nodes over several GPUs from this process) excludes distributed, node_range and device")
    if method.distributed and method.node_range is not None:
        raise ValueError("HIP: distributed=True derives the node range from the rank; node_range must not be given")
    if method.precision not in _lib.PRECISIONS:
        raise ValueError(f"HIP: unknown precision {method.precision!r} (use 'auto', 'i8x', 'i8w' or 'f64')")
    world, rank = 1, 0
    if method.distributed:
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
    node_range = method.node_range or _node_partition(n, world, rank)
    device = method.device
    if device is None:
        device = 0
        if method.distributed:
            import torch
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    solve = _local_solve_hip  # (module attribute, looked up per call: the CPU-only tests of this layer substitute the oracle)
    try:
        if method.devices is not None:
            out, kkt, st, keys = _local_solve_multi(samples, formulation, method, order)
            node_range = (0, n)
        else:
            out, kkt, st, keys = solve(samples, formulation, method, order, node_range, device)
    except _lib.GMLConvergenceError as e:  # the reference's @assert (:180): keep what the solver reached
        method.stats.clear()
        method.stats.update(getattr(e, "stats", {}) or {})
        method.stats["kkt"] = getattr(e, "kkt", None)
        raise
    method.stats.clear()
    method.stats.update(st or {})
    method.stats["kkt"] = kkt

    if method.distributed and world > 1:
        # the ranks may share a GPU with each other and share it with torch's allocator (the gather below): hand the blocks the
        # library keeps for the next handle back to the driver before anybody else needs the memory
        if solve is _local_solve_hip:
            _lib.trim_cache()
        P = out.shape[1]
        out = _gather_rows(out, n, P, method)
        node_range = (0, n)

    if isinstance(formulation, multiRISE):
        if keys is None or method.distributed and world > 1:
            keys = _all_multi_keys(n, order, node_range, pairwise_slots=(order == 2 and getattr(solve, "pairwise_slots", False)))
        rec = {}
        for r, u in enumerate(range(node_range[0], node_range[1])):
            for key, v in zip(keys[r], out[r]):
                rec[tuple(i + 1 for i in key)] = float(v)  # (u, ascending others), 1-based (:129-132)
        if formulation.symmetrization:  # :135-149 group by sorted key, mean
            groups = {}
            for k, v in rec.items():
                groups.setdefault(tuple(sorted(k)), []).append(v)
            rec = {k: float(np.mean(v)) for k, v in groups.items()}
        return FactorGraph(order, n, "spin", rec)  # :151

    R = np.array(out)  # rows node_range; out[u, :] = reconstruction[u, 1:n] (:181)
    if formulation.symmetrization and R.shape[0] == R.shape[1]:
        R = 0.5 * (R + R.T)  # :184-186
    return R


def _all_multi_keys(n, order, node_range, pairwise_slots=False):
    """(u), (u,i)..., (u,i,j)... in the reference's order (:94-104, models.jl:228-246); 0-based.
    pairwise_slots: the order-2 slot layout of the C ABI (slot i <-> spin i, slot u = field)."""
    from itertools import combinations
    keys = []
    if pairwise_slots:
        return [[(u,) if i == u else (u, i) for i in range(n)] for u in range(node_range[0], node_range[1])]
    for u in range(node_range[0], node_range[1]):
        others = [i for i in range(n) if i != u]
        ku = []
        for p in range(1, order + 1):
            ku.extend((u,) + c for c in combinations(others, p - 1))
        keys.append(ku)
    return keys
