"""CPU oracle for the learn() hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this
package; the product path (graphicalmodellearning.jl_amd / libgml_hip.so) never does.
"""
from .oracle import *  # noqa: F401,F403
