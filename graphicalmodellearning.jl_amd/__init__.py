"""MI355X-native learn() hot path of GraphicalModelLearning.jl.

Same surface as the reference module (GraphicalModelLearning.jl:3-6, models.jl:3): learn,
the GMLFormulation types, the GMLMethod types (NLP plus the new HIP), FactorGraph.  The compute
path is libgml_hip.so (hand-written HIP kernels for gfx950 behind the C ABI in include/gml.h).
"""
from ._lib import GMLConvergenceError, GMLError, MultiProblem, Problem, lib  # noqa: F401
from .factor_graph import FactorGraph, matrix_to_terms, permutations, check_model_data  # noqa: F401
from .formulations import (HIP, ISODUS, NLP, RISE, RISEA, RPLE, GMLFormulation, GMLMethod,  # noqa: F401
                           logRISE, multiRISE)
from .learn import learn  # noqa: F401
from .sampling import GMSampler, Gibbs, Glauber, sample  # noqa: F401

__all__ = ["learn", "GMLFormulation", "RISE", "logRISE", "RPLE", "RISEA", "multiRISE", "ISODUS", "GMLMethod",
           "NLP", "HIP", "FactorGraph", "Problem", "MultiProblem", "GMLError", "GMLConvergenceError", "sample", "GMSampler", "Gibbs"]
