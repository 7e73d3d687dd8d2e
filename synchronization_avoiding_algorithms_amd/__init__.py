"""MI355X-native hot path of desResLab/Synchronization-avoiding-algorithms.

Explicit linear-tetrahedral elastodynamics (fused HIP kernels behind the C ABI of
``include/saa_hip.h``) plus the LSTM shared-node predictor on PyTorch-ROCm.  See DESIGN.md.
"""
from . import hip_graphs

hip_graphs.configure_runtime()  # before anything can initialise HIP: see hip_graphs.py (captured graphs with reductions)

from . import _lib, fem_setup, mesh  # noqa: F401,E402
from .solver import HipExplicitSolver, plan_host_stats  # noqa: F401,E402

__all__ = ["HipExplicitSolver", "plan_host_stats", "mesh", "fem_setup"]
