"""Counterpart of the reference's ``Tools/Shape_function_Deriv.py``: Lagrange shape functions of the 4-node (``p = 1``) and
10-node (``p = 2``) tetrahedron, their parametric derivatives, the Jacobian of the isoparametric map and the map itself.
The dynamic path uses ``p = 1`` only (``Data_prepare.py:43-44``); on the GPU these quantities never exist as such - the
step kernels fold them into the closed-form element force (``csrc/saa_kernels.hip: tet_core``).  The functions are here
for driver code and post-processing written against the reference's names.

Both degrees are written in barycentric coordinates ``L = (1 - xi - eta - zeta, xi, eta, zeta)``: vertex functions
``L_a`` (p = 1) or ``L_a (2 L_a - 1)`` (p = 2), edge functions ``4 L_a L_b`` in the reference's edge order
(0-1, 1-2, 0-2, 0-3, 1-3, 2-3; ``Shape_function_Deriv.py:13-23``)."""
from __future__ import annotations

import numpy as np

#: d L_a / d xi_j : node 0 carries 1 - xi - eta - zeta, node a > 0 the a-th coordinate
_DL = np.vstack([-np.ones((1, 3)), np.eye(3)])
#: mid-side nodes 4..9 of the 10-node tet sit on these edges (Shape_function_Deriv.py:17-22)
_EDGES = np.array([(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)])


def _bary(xi):
    xi = np.asarray(xi, dtype=np.float64).reshape(-1)[:3]
    return np.concatenate([[1.0 - xi[0] - xi[1] - xi[2]], xi])


def _degree(p):
    if p not in (1, 2):
        raise NotImplementedError("Lagrange tetrahedra of degree 1 and 2 (what the reference provides)")


def Shape_Function(p, xi):
    """Shape functions at ``xi`` -> ``(4,)`` for p = 1, ``(10,)`` for p = 2 (``Shape_function_Deriv.py:9-23``)."""
    _degree(p)
    L = _bary(xi)
    if p == 1:
        return L
    return np.concatenate([L * (2.0 * L - 1.0), 4.0 * L[_EDGES[:, 0]] * L[_EDGES[:, 1]]])


def Shape_Deri(p, xi):
    """Parametric derivatives ``(4,3)`` (constant) or ``(10,3)`` (``Shape_function_Deriv.py:33-47``)."""
    _degree(p)
    if p == 1:
        return _DL.copy()
    L = _bary(xi)
    vertex = (4.0 * L - 1.0)[:, None] * _DL                                       # d/dxi of L_a (2 L_a - 1)
    edge = 4.0 * (L[_EDGES[:, 1]][:, None] * _DL[_EDGES[:, 0]] + L[_EDGES[:, 0]][:, None] * _DL[_EDGES[:, 1]])
    return np.vstack([vertex, edge])


def Jacobian(p, P, local_xi):
    """``J[i,j] = sum_a dN_a/dxi_j * P[a,i]`` (``Shape_function_Deriv.py:60-67``); for p = 1 its columns are the edges
    ``x_a - x_0``.  ``P``: the element's 4 or 10 nodes, one per row."""
    _degree(p)
    n = 4 if p == 1 else 10
    return np.asarray(P, dtype=np.float64)[:n].T @ Shape_Deri(p, local_xi)


def IsoparametricMap(p, P, local_xi):
    """Physical point of ``local_xi`` as a ``(3,1)`` column (``Shape_function_Deriv.py:77-82``)."""
    _degree(p)
    n = 4 if p == 1 else 10
    return (np.asarray(P, dtype=np.float64)[:n].T @ Shape_Function(p, local_xi)).reshape(3, 1)
