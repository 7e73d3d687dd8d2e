"""Counterpart of the reference's ``Tools/Dynamic_solver.py``."""
from __future__ import annotations

import numpy as np

from .Distributed_tools import syn_cpus


def parallel_explicit_solver_dis_pre(LocalK, F_rankwise, Points, Local_nodes, Local_Dirichlet, T, Elas, l_M, alpha,
                                     size, rank, MODEL=False):
    """One explicit step with the reference's signature and semantics (``Dynamic_solver.py:9-34``):
    ``F_int = LocalK.dot(T.d0)`` on the GPU, the damped central-difference update in the reference's
    association order (also on the GPU, bit-identical to NumPy), Dirichlet zeroing; if ``MODEL`` is
    false and ``size != 1`` the internal forces are summed over ranks first (``syn_cpus``).  Pure
    function of its inputs; returns a new ``(3n,1)`` array.

    This is the compatibility path (state crosses PCIe every call).  The device-resident fast path is
    ``HipExplicitSolver.step`` / ``PartitionedSolver``.
    """
    F_int = LocalK.dot(T.d0)
    if not MODEL and size != 1:
        F_int = syn_cpus(size, rank, F_int, len(Points), Local_nodes)
    if hasattr(LocalK, "explicit_update"):
        return LocalK.explicit_update(F_int, F_rankwise, l_M, Local_Dirichlet, T, alpha)
    # any other operator with .dot (e.g. a scipy matrix): same expression on the host
    from .commons import linear_ramp

    F_ext = (np.asarray(F_rankwise) * linear_ramp(T.tn)).reshape((len(F_rankwise), 1))
    l_M = np.asarray(l_M).reshape((len(l_M), 1))
    d1 = (T.dt ** 2 * (F_ext - F_int) + 2 * l_M * T.d0 - l_M * T.dn + T.dt / 2 * l_M * alpha * T.dn) / (
        l_M + 0.5 * alpha * l_M * T.dt)
    d1[Local_Dirichlet] = 0
    return d1
