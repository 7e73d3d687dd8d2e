"""Counterpart of the reference's ``Tools/Qudrature.py`` (the module name keeps the reference's spelling).

The explicit path integrates with ``Gauss_Legendre(2)`` - four points, exact for the constant integrands of a linear
tet (``Mat_construction.py:84-87``); on the GPU the rule is folded into the closed form ``K_e = (detJ/6) B^T D B``
(weights sum to 1/6).  The 5-point degree-3 rule is provided as well; higher rules are not on this path."""
from __future__ import annotations

import numpy as np


def Gauss_Legendre(n):
    """``(nodes (nq,3), weights (nq,))`` on the reference tetrahedron (volume 1/6), ``Qudrature.py:6-45``."""
    if n == 2:
        # symmetric 4-point rule: one barycentric coordinate a = (5 + 3 sqrt 5)/20, the others b = (5 - sqrt 5)/20
        a, b = 0.5854101966249685, 0.1381966011250105
        nodes = np.full((4, 3), b)
        nodes[[0, 1, 2], [0, 1, 2]] = a
        return nodes, np.full(4, 0.25 / 6)
    if n == 3:
        # centroid with negative weight + four points at barycentric (1/2, 1/6, 1/6, 1/6)
        nodes = np.full((5, 3), 1.0 / 6)
        nodes[0] = 1.0 / 4
        nodes[[1, 2, 3], [0, 1, 2]] = 1.0 / 2
        return nodes, np.array([-4.0 / 5 / 6] + [9.0 / 20 / 6] * 4)
    raise NotImplementedError("rules beyond n = 3 are not used by the explicit path (Mat_construction.py:84-87)")
