"""Counterpart of the reference's ``Tools/Steady_solvers.py`` (steady solve only).

``Steady_Elasticity_solver`` keeps the reference's signature (``Steady_solvers.py:13``) and returns the same
``(3N,1)`` array, but never forms the dense ``(3N)^2`` matrix of ``Global_Assembly``: the system is solved by
preconditioned conjugate gradients on the GPU with the matrix-free element kernel as the operator."""
from __future__ import annotations

import numpy as np

from ..fem_setup import device_setup_fields
from ..solver import HipExplicitSolver
from ..steady import steady_solve, stiffness_diagonal


def Steady_Elasticity_solver(p, Cells, Points, Dirichlet, elas, t=None, Facets=None, Neumann=None, device=0,
                             tol=1e-12):
    """Solve ``K d = F`` with ``d[Dirichlet] = 0`` (``Steady_solvers.py:13-22``).  ``Cells`` hold global node ids,
    ``Dirichlet`` global dofs (``node_to_dof``), ``elas`` the un-ramped ``elasticity`` object
    (``Data_prepare.py:161``); ``p`` must be 1 here (linear tetrahedra)."""
    if p != 1:
        raise NotImplementedError("linear tetrahedra only on the GPU path")
    if Neumann is not None or Facets is not None:
        raise NotImplementedError("the reference passes Facets=None, Neumann=None (Data_prepare.py:163)")
    Points = np.ascontiguousarray(Points, dtype=np.float64)
    Cells = np.ascontiguousarray(np.asarray(Cells)[:, :4], dtype=np.int32)
    scale = 1.0
    if getattr(elas, "R", False) and t is not None:  # ramped load evaluated at time t (commons.py:35-41)
        scale = t if t <= 1 else 1.0
    lumped, load, _ = device_setup_fields(Points, Cells, elas.rho, elas.fz * scale, device)
    dirichlet = np.asarray(sorted(Dirichlet), dtype=np.int32)
    sol = HipExplicitSolver(Points, Cells, lumped, load, dirichlet, elas.lmd, elas.mu, 1.0, 0.0, device=device)
    try:
        d, _, _ = steady_solve(sol, load, dirichlet, diag=stiffness_diagonal(Points, Cells, elas.lmd, elas.mu), tol=tol)
    finally:
        sol.close()
    return d
