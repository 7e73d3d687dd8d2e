"""MI355X-native hot path of desResLab/Synchronization-avoiding-algorithms.

Explicit linear-tetrahedral elastodynamics (fused HIP kernels behind the C ABI of
``include/saa_hip.h``) plus the LSTM shared-node predictor on PyTorch-ROCm.  See DESIGN.md.
"""
from . import _lib, fem_setup, mesh  # noqa: F401
from .solver import HipExplicitSolver, plan_host_stats  # noqa: F401

__all__ = ["HipExplicitSolver", "plan_host_stats", "mesh", "fem_setup"]
