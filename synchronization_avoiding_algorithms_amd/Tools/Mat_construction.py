"""Counterpart of the reference's ``Tools/Mat_construction.py`` for the explicit path.

``Local_assembly_for_stiffness`` returns an operator whose ``.dot`` runs the matrix-free HIP element
kernel - the reference's only use of ``LocalK`` (``Dynamic_solver.py:12``) - instead of a dense
``(3n)^2`` array turned CSR (``Mat_construction.py:122-150``)."""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix

from ..fem_setup import element_stiffness, signed_volumes
from ..solver import HipExplicitSolver
from .Distributed_tools import local_mat_node


class HipStiffnessOperator:
    """``LocalK``: ``.dot(d)`` = ``K_local @ d`` for ``d (3n,1)`` float64, evaluated on the GPU."""

    def __init__(self, points_local, cells_local, lmd, mu, device=0):
        self._pts = np.ascontiguousarray(points_local, dtype=np.float64)
        self._cells = np.ascontiguousarray(cells_local, dtype=np.int32)
        self.lmd, self.mu, self.device = float(lmd), float(mu), device
        n = len(self._pts)
        self.shape = (3 * n, 3 * n)
        self._solver = None
        self._key = None

    def _ensure(self, l_M=None, F=None, dirichlet=(), dt=1.0, alpha=0.0):
        n3 = self.shape[0]
        l_M = np.ones(n3) if l_M is None else np.asarray(l_M, dtype=np.float64).reshape(-1)
        F = np.zeros(n3) if F is None else np.asarray(F, dtype=np.float64).reshape(-1)
        dirichlet = np.asarray(dirichlet, dtype=np.int32).reshape(-1)
        key = (float(dt), float(alpha), dirichlet.tobytes())
        if self._solver is None or key != self._key:
            if self._solver is not None:
                self._solver.close()
            self._solver = HipExplicitSolver(self._pts, self._cells, l_M, F, dirichlet, self.lmd, self.mu,
                                             float(dt), float(alpha), device=self.device)
            self._key, self._lm, self._f = key, l_M.copy(), F.copy()
        elif not (np.array_equal(l_M, self._lm) and np.array_equal(F, self._f)):
            self._solver.set_loads(F, l_M)
            self._lm, self._f = l_M.copy(), F.copy()
        return self._solver

    def dot(self, d):
        return self._ensure().internal_force(d) if self._solver is None else self._solver.internal_force(d)

    def explicit_update(self, F_int, F_rankwise, l_M, dirichlet, T, alpha):
        """``Dynamic_solver.py:13-20`` on the GPU, bit-identical to the NumPy expression."""
        return self._ensure(l_M, F_rankwise, dirichlet, T.dt, alpha).cd_update(F_int, T.d0, T.dn, T.tn)


def Local_assembly_for_stiffness(local_node_list, Cell, Points, deg, n_basis, elas, rank, device=0):
    """Same signature as ``Mat_construction.py:122``.  ``Cell`` rows hold GLOBAL node ids of this rank's
    elements, ``local_node_list`` the rank's nodes; only ``deg == 1`` (4-node tets) is dynamic-capable in
    the reference too (``Data_prepare.py:43-44``)."""
    if deg != 1 or n_basis != 4:
        raise NotImplementedError("the explicit path uses linear tetrahedra only (Data_prepare.py:43-44)")
    Points = np.asarray(Points, dtype=np.float64)
    nodes = np.asarray(local_node_list, dtype=np.int64)
    cells = np.asarray(Cell, dtype=np.int64)
    local = np.asarray(local_mat_node(cells.ravel(), nodes), dtype=np.int32).reshape(-1, 4)
    return HipStiffnessOperator(Points[nodes], local, elas.lmd, elas.mu, device=device)


def Global_Assembly_no_bc(deg, Cells, Points, elas, t):
    """``Mat_construction.py:199-231`` for its one use (``Data_prepare.py:175-176``): the consistent mass
    whose row sums give the lumped mass, and the pre-assembled body force.  Returns ``(M, None, F)`` with
    ``M`` scipy-CSR (the dense K of the reference is never used there and is not formed)."""
    if deg != 1:
        raise NotImplementedError("linear tetrahedra only")
    Points = np.asarray(Points, dtype=np.float64)
    Cells = np.asarray(Cells, dtype=np.int64)
    n = len(Points)
    vol = signed_volumes(Points, Cells)
    # consistent mass of a linear tet: rho*V/20*(1 + delta_ab), the same on each axis
    blk = (np.ones((4, 4)) + np.eye(4)) / 20.0
    vals = (elas.rho * vol)[:, None, None] * blk[None]
    rows = np.repeat(Cells, 4, axis=1).reshape(-1, 4, 4)
    cols = np.tile(Cells, (1, 4)).reshape(-1, 4, 4)
    M = csr_matrix((3 * n, 3 * n))
    for A in range(3):
        M = M + csr_matrix((vals.ravel(), (3 * rows.ravel() + A, 3 * cols.ravel() + A)), shape=(3 * n, 3 * n))
    f = elas.f(None, t).ravel()
    nodal = np.bincount(Cells.ravel(), weights=np.repeat(vol / 4.0, 4), minlength=n)
    F = np.outer(nodal, f).reshape(-1, 1)
    return M, None, F


#: (3N) beyond which ``Global_Assembly`` hands back scipy-CSR matrices instead of dense arrays
DENSE_DOF_LIMIT = 12000


def Global_Assembly(deg, Cells, Points, Dirichlet, elas, t, Facets=None, Neumann=None, steady=False, sparse=None):
    """``Mat_construction.py:155-196``: consistent mass, stiffness and load with the rows and columns of the Dirichlet
    dofs left out (the caller then puts 1 on their diagonal: ``Data_prepare.py:185-190``, ``Steady_solvers.py:16-21``).
    Same call, same return triple ``(M, K, F)`` with ``F (3N,1)``.

    Assembled from closed-form element matrices in O(N) instead of the reference's Python quadruple loop over dense
    ``(3N)^2`` arrays.  ``M`` and ``K`` are dense ``ndarray`` (what ``np.linalg.solve`` in the reference's drivers needs)
    while ``3N <= DENSE_DOF_LIMIT``, scipy CSR beyond it (a dense 1M-tet matrix would hold 2.6 TB); ``sparse=True/False``
    forces either.  ``steady`` only changes how prescribed NON-zero Dirichlet values would enter ``F`` in the reference,
    and those are all zero there (``:187-190``), so it has no effect here either."""
    if deg != 1:
        raise NotImplementedError("linear tetrahedra only (Data_prepare.py:43-44)")
    Points = np.asarray(Points, dtype=np.float64)
    Cells = np.asarray(Cells, dtype=np.int64)[:, :4]
    n3 = 3 * len(Points)
    free = np.ones(n3, dtype=bool)
    free[np.asarray(list(Dirichlet), dtype=np.int64)] = False
    vol = signed_volumes(Points, Cells)
    dof = (3 * Cells[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 12)
    rows, cols = np.repeat(dof, 12, axis=1).ravel(), np.tile(dof, (1, 12)).ravel()
    keep = free[rows] & free[cols]
    Ke = element_stiffness(Points, Cells, elas.lmd, elas.mu)
    # consistent mass rho*V/20*(1 + delta_ab) on each axis (the 4-point rule integrates N_a N_b exactly)
    Me = np.kron((np.ones((4, 4)) + np.eye(4)) / 20.0, np.eye(3))[None] * (elas.rho * vol)[:, None, None]
    K = csr_matrix((Ke.ravel()[keep], (rows[keep], cols[keep])), shape=(n3, n3))
    M = csr_matrix((Me.ravel()[keep], (rows[keep], cols[keep])), shape=(n3, n3))
    nodal = np.bincount(Cells.ravel(), weights=np.repeat(vol / 4.0, 4), minlength=len(Points))
    F = np.outer(nodal, np.asarray(elas.f(None, t), dtype=np.float64).ravel()).reshape(-1, 1)
    F[~free] = 0.0
    if sparse is None:
        sparse = n3 > DENSE_DOF_LIMIT
    return (M, K, F) if sparse else (M.toarray(), K.toarray(), F)
