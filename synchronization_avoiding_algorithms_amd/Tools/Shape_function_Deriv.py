"""Counterpart of the reference's ``Tools/Shape_function_Deriv.py`` for linear tetrahedra (``p = 1``), the only
degree the dynamic path uses (``Data_prepare.py:43-44``).  On the GPU these quantities never exist as such - the
step kernels fold them into the closed-form element force (``csrc/saa_kernels.hip: tet_forces``); the functions are
here for driver code and post-processing written against the reference's names."""
from __future__ import annotations

import numpy as np

#: d N_a / d xi_j of the 4-node tet: node 0 carries 1 - xi - eta - zeta, node a > 0 the a-th coordinate
_DN = np.vstack([-np.ones((1, 3)), np.eye(3)])


def _p1(p):
    if p != 1:
        raise NotImplementedError("linear tetrahedra only (the reference's dynamics are p = 1 too, Data_prepare.py:43-44)")


def Shape_Function(p, xi):
    """Barycentric shape functions at ``xi`` -> ``(4,)`` (``Shape_function_Deriv.py:9-12``)."""
    _p1(p)
    xi = np.asarray(xi, dtype=np.float64).reshape(-1)[:3]
    return np.concatenate([[1.0 - xi[0] - xi[1] - xi[2]], xi])


def Shape_Deri(p, xi):
    """Parametric derivatives ``(4,3)``, constant for p = 1 (``Shape_function_Deriv.py:33-36``)."""
    _p1(p)
    return _DN.copy()


def Jacobian(p, P, local_xi):
    """``J[i,j] = sum_a dN_a/dxi_j * P[a,i]`` (``Shape_function_Deriv.py:60-67``): columns are the edges ``x_a - x_0``."""
    _p1(p)
    return np.asarray(P, dtype=np.float64)[:4].T @ _DN


def IsoparametricMap(p, P, local_xi):
    """Physical point of ``local_xi`` as a ``(3,1)`` column (``Shape_function_Deriv.py:77-82``)."""
    _p1(p)
    return (np.asarray(P, dtype=np.float64)[:4].T @ Shape_Function(p, local_xi)).reshape(3, 1)
