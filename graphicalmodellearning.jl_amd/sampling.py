"""sample(model, N): host mirror of the reference's sampler front door (src/sampling.jl:90-106) on top of
the on-device exact sampler (gml_problem_create_sampled / _sampled_terms).  Any interaction order; every
connected component of the term hypergraph must have at most 22 spins (the reference enumerates all 2^n
states of the whole model, which limits it to n ~ 25)."""
import numpy as np

from . import _lib
from .factor_graph import FactorGraph


class GMSampler:  # sampling.jl:7
    pass


class Gibbs(GMSampler):  # sampling.jl:9  (the reference's "Gibbs" sampler is exact enumeration, as is this one)
    pass


class Glauber(GMSampler):
    """Not in the reference: N independent heat-bath chains of `sweeps` sweeps on the device, for models whose
    connected components exceed the 22 spins exact enumeration can handle (gml_problem_create_mcmc_terms)."""

    def __init__(self, sweeps=200):
        self.sweeps = int(sweeps)


def _problem_args(model):
    """Keyword arguments of _lib.Problem for a model: matrix (order <= 2, :98-99) or term list (:100-101)."""
    if isinstance(model, FactorGraph):
        if model.alphabet != "spin":
            raise ValueError(f"sampling is only supported for spin FactorGraphs, given alphabet {model.alphabet}")  # :97
        if model.order <= 2:
            return {"model": model.to_matrix()}
        return {"terms": model.terms, "n": model.varible_count, "order": model.order}
    if isinstance(model, dict):
        return {"terms": model, "order": max(2, max(len(k) for k in model))}
    return {"model": np.asarray(model, dtype=np.float64)}


def sample(model, number_sample, replicates=None, sampler=None, *, seed=0, device=0):
    """sample(gm, N) -> histogram matrix [count, s_1..s_n], one row per observed configuration
    (sampling.jl:52-54); sample(gm, N, replicates) -> list of such matrices (:91)."""
    args = _problem_args(model)
    if isinstance(sampler, Glauber):
        if "model" in args:  # the chains run on term lists
            fg = model if isinstance(model, FactorGraph) else FactorGraph(np.asarray(model, dtype=np.float64))
            args = {"terms": fg.terms, "n": fg.varible_count, "order": 2}
        args["mcmc_sweeps"] = sampler.sweeps
    reps = 1 if replicates is None else int(replicates)
    nspins = args["model"].shape[0] if "model" in args else args["n"] if "n" in args else max(max(k) for k in args["terms"] if len(k))
    out = []
    for b in range(reps):
        kw = dict(num_samples=int(number_sample), seed=int(seed) + 7919 * b, device=device, **args)
        if nspins <= 64 and int(number_sample) < 2 ** 31:
            # countmap (sampling.jl:52) on the device: the draws are sorted and run-length encoded there; only the distinct
            # configurations and their counts come back
            with _lib.Problem(histogram=True, **kw) as p:
                states, counts = p.spins(), p.counts()
            order = np.lexsort(states.T[::-1])  # rows in the order np.unique would give them
            states, counts = states[order], counts[order]
        else:
            with _lib.Problem(**kw) as p:
                spins = p.spins()
            states, counts = np.unique(spins, axis=0, return_counts=True)
        out.append(np.concatenate([np.rint(counts)[:, None].astype(np.int64), states.astype(np.int64)], axis=1))
    return out[0] if replicates is None else out
