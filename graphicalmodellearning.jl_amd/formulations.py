"""The reference's formulation / method types (GraphicalModelLearning.jl:20-65), same names,
same field order, same defaults -- plus the one new GMLMethod subtype, HIP."""
from dataclasses import dataclass, field
from typing import Any, Optional, Sequence, Tuple


class GMLFormulation:  # :20
    pass


@dataclass
class multiRISE(GMLFormulation):  # :22-28
    regularizer: float = 0.4
    symmetrization: bool = True
    interaction_order: int = 2


# BASELINE.json calls the multi-body estimator "ISODUS"; the reference's name is multiRISE.
ISODUS = multiRISE


@dataclass
class RISE(GMLFormulation):  # :30-35
    regularizer: float = 0.4
    symmetrization: bool = True


@dataclass
class RISEA(GMLFormulation):  # :37-42 (same objective as RISE, hand-written obj/grad :191-208)
    regularizer: float = 0.4
    symmetrization: bool = True


@dataclass
class logRISE(GMLFormulation):  # :44-49
    regularizer: float = 0.8
    symmetrization: bool = True


@dataclass
class RPLE(GMLFormulation):  # :51-56
    regularizer: float = 0.2
    symmetrization: bool = True


class GMLMethod:  # :59
    pass


@dataclass
class HIP(GMLMethod):
    """Solve every node-wise problem on MI355X through libgml_hip (include/gml.h).

    tol        KKT tolerance (max |pseudo-gradient| per node)
    precision  "i8x" (int8-limb fixed point on the i8 MFMA: the fast path; rows it cannot bring below tol are finished
               on the FP64 path unless polish=False), "i8w" (the same int8 matrix cores at the width of the reference's
               Float64 arithmetic: theta in 54-bit, the weights in 47-bit limbs, exp in FP64 -- objective and gradient to the
               1e-12 the FP64 path is held to, ~1.4x the time of "i8x"), "f64" (FP64 MFMA throughout) or "auto" (the
               default: "i8x"; "i8w" for tolerances below 2e-10 and for small problems -- the reference's fixtures, the README example)
    device     HIP device ordinal; with distributed=True the local rank's device
    devices    several GPUs of this node from this one process: the library shards the nodes over them (one host thread
               per GPU, gml_multi_*); what the Julia wrapper's HIP(devices = 0:7) binds
    distributed  shard the nodes over torch.distributed ranks (one process per GPU) and gather the rows (RCCL)
    """
    tol: float = 1e-9
    precision: str = "auto"
    device: Optional[int] = None
    devices: Optional[Sequence[int]] = None
    max_iter: int = 100
    max_working: int = 512
    max_add: int = 64
    hess_samples: int = 0  # configurations per Newton Hessian: 0 = adaptive (32768 x rows / active rows), < 0 = all
    polish: bool = True
    verbose: int = 0
    distributed: bool = False
    node_range: Optional[Tuple[int, int]] = None
    stats: dict = field(default_factory=dict, repr=False, compare=False)


@dataclass
class NLP(GMLMethod):  # :61-65
    """The reference's method type: `solver` is a JuMP optimizer factory there.  Neither Julia
    nor Ipopt exists on this platform, so NLP(...) is accepted for source compatibility and
    dispatches to the same device solver as HIP() with default options."""
    solver: Any = None
