"""sample(model, N): host mirror of the reference's sampler front door (src/sampling.jl:90-106) on top of
the on-device exact sampler (gml_problem_create_sampled).  Pairwise models only; every connected component
of the coupling graph must have at most 22 spins (the reference enumerates all 2^n states of the whole
model, which limits it to n ~ 25)."""
import numpy as np

from . import _lib
from .factor_graph import FactorGraph


class GMSampler:  # sampling.jl:7
    pass


class Gibbs(GMSampler):  # sampling.jl:9  (the reference's "Gibbs" sampler is exact enumeration, as is this one)
    pass


def _as_matrix(model):
    if isinstance(model, FactorGraph):
        if model.alphabet != "spin":
            raise ValueError(f"sampling is only supported for spin FactorGraphs, given alphabet {model.alphabet}")  # :97
        return model.to_matrix()
    return np.asarray(model, dtype=np.float64)


def sample(model, number_sample, replicates=None, sampler=None, *, seed=0, device=0):
    """sample(gm, N) -> histogram matrix [count, s_1..s_n], one row per observed configuration
    (sampling.jl:52-54); sample(gm, N, replicates) -> list of such matrices (:91)."""
    m = _as_matrix(model)
    reps = 1 if replicates is None else int(replicates)
    out = []
    for b in range(reps):
        with _lib.Problem(model=m, num_samples=int(number_sample), seed=int(seed) + 7919 * b, device=device) as p:
            spins = p.spins()
        states, counts = np.unique(spins, axis=0, return_counts=True)  # countmap (sampling.jl:52)
        out.append(np.concatenate([counts[:, None].astype(np.int64), states.astype(np.int64)], axis=1))
    return out[0] if replicates is None else out
