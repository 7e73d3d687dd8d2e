"""Steady state ``K d = F`` on the GPU, matrix-free.

The reference solves it once before the time stepping, on rank 0, with a dense ``np.linalg.solve`` on the
``(3N)^2`` matrix of ``Global_Assembly`` (/root/reference ``Tools/Steady_solvers.py:13-22``, called at
``Data_prepare.py:157-168`` with the un-ramped load) and writes ``Results/Static/steady_distributed.vtk``.
Here: Jacobi-preconditioned conjugate gradients whose operator is the FORCE_ONLY kernel of the time stepper
(``saa_internal_force_device``), all vectors resident on the GPU as float64 torch tensors; Dirichlet dofs are
eliminated the way ``Global_Assembly`` does it (rows and columns dropped, ``Mat_construction.py:176-192``).
"""
from __future__ import annotations

import numpy as np


def stiffness_diagonal(points, cells, lmd, mu) -> np.ndarray:
    """diag(K) of the linear-tet stiffness, ``(3N,)``: ``V (lambda g_A^2 + mu (|g|^2 + g_A^2))`` per node ``a`` and
    component ``A`` of every element, ``g = grad N_a`` (closed form of ``Local_K_coronary``,
    ``Mat_construction.py:79-119``)."""
    p = np.asarray(points, dtype=np.float64)[np.asarray(cells)]
    e1, e2, e3 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]
    c1, c2, c3 = np.cross(e2, e3), np.cross(e3, e1), np.cross(e1, e2)
    det = np.einsum("ij,ij->i", e1, c1)
    g = np.stack([-(c1 + c2 + c3), c1, c2, c3], axis=1) / det[:, None, None]          # (Ne, 4, 3)
    vol = (det / 6.0)[:, None, None]
    diag_e = vol * (lmd * g ** 2 + mu * ((g ** 2).sum(axis=2, keepdims=True) + g ** 2))  # (Ne, 4, 3)
    dof = 3 * np.asarray(cells)[:, :, None] + np.arange(3)[None, None, :]
    return np.bincount(dof.ravel(), weights=diag_e.ravel(), minlength=3 * len(points))


def steady_solve(solver, f_ext, dirichlet_dofs, diag=None, tol=1e-12, max_iter=None, check_every=25, device=None):
    """Solve ``K d = f_ext`` with ``d[dirichlet_dofs] = 0`` -> ``(d (3n,1) numpy, iterations, relative residual)``.

    ``solver``: a :class:`HipExplicitSolver` (caller numbering); ``diag``: optional ``diag(K)`` for the Jacobi
    preconditioner (:func:`stiffness_diagonal`).  Converged when ``|r| <= tol |b|``.
    """
    import torch

    dev = torch.device("cuda", solver.device if device is None else device)
    solver.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    n = solver.n_dof
    free = torch.ones(n, dtype=torch.float64, device=dev)
    if len(dirichlet_dofs):
        free[torch.as_tensor(np.asarray(dirichlet_dofs, dtype=np.int64), device=dev)] = 0.0
    b = torch.as_tensor(np.asarray(f_ext, dtype=np.float64).reshape(-1), device=dev) * free
    minv = free.clone()
    if diag is not None:
        dk = torch.as_tensor(np.asarray(diag, dtype=np.float64).reshape(-1), device=dev)
        minv = torch.where(dk > 0, free / dk, free)
    x = torch.zeros_like(b)
    r = b.clone()
    z = minv * r
    p = z.clone()
    ap = torch.empty_like(b)
    rz = torch.dot(r, z)
    bnorm = float(torch.linalg.vector_norm(b))
    if bnorm == 0.0:
        return np.zeros((n, 1)), 0, 0.0
    max_iter = max_iter or 20 * n
    it, rel = 0, 1.0
    while it < max_iter:
        solver.internal_force_device(p, ap)      # K p on the GPU (no matrix anywhere)
        ap.mul_(free)
        alpha = rz / torch.dot(p, ap)
        x.add_(alpha * p)
        r.sub_(alpha * ap)
        z = minv * r
        rz_new = torch.dot(r, z)
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
        if it % check_every == 0:                # the only host synchronisation
            rel = float(torch.linalg.vector_norm(r)) / bnorm
            if rel <= tol:
                break
    rel = float(torch.linalg.vector_norm(r)) / bnorm
    return x.cpu().numpy().reshape(-1, 1), it, rel


def write_vtk_point_data(path, points, cells, displacement):
    """Legacy-VTK file with the mesh and the point data ``displacement-x/-y/-z`` - the arrays the reference hands to
    ``meshio.write_points_cells`` at ``Data_prepare.py:165-168`` (ASCII here)."""
    import os

    points, cells = np.asarray(points, dtype=np.float64), np.asarray(cells)
    d = np.asarray(displacement, dtype=np.float64).reshape(-1, 3)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as fh:
        fh.write("# vtk DataFile Version 4.2\nsteady solution d = K^-1 F\nASCII\nDATASET UNSTRUCTURED_GRID\n")
        fh.write(f"POINTS {len(points)} double\n")
        np.savetxt(fh, points, fmt="%.17g")
        fh.write(f"CELLS {len(cells)} {5 * len(cells)}\n")
        np.savetxt(fh, np.column_stack([np.full(len(cells), 4), cells]), fmt="%d")
        fh.write(f"CELL_TYPES {len(cells)}\n")
        np.savetxt(fh, np.full(len(cells), 10), fmt="%d")
        fh.write(f"POINT_DATA {len(points)}\n")
        for c, name in enumerate(("displacement-x", "displacement-y", "displacement-z")):
            fh.write(f"SCALARS {name} double 1\nLOOKUP_TABLE default\n")
            np.savetxt(fh, d[:, c], fmt="%.17g")
    return path
