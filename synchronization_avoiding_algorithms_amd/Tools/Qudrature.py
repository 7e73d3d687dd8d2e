"""Counterpart of the reference's ``Tools/Qudrature.py`` (the module name keeps the reference's spelling).

The explicit path integrates with ``Gauss_Legendre(2)`` - four points, exact for the constant integrands of a linear
tet (``Mat_construction.py:84-87``); on the GPU the rule is folded into the closed form ``K_e = (detJ/6) B^T D B``
(weights sum to 1/6).  The 5-point degree-3 rule and the 14-point degree-4 rule of the reference are provided as well,
each built from its symmetry orbits in barycentric coordinates."""
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
    if n == 4:
        # 14 points (Qudrature.py:21-45): the six edge mid-points, and two orbits of four points at barycentric
        # (1 - 3b, b, b, b) - listed with the heavy coordinate on x, on node 0, on z, on y, as the reference lists them
        nodes = np.empty((14, 3))
        nodes[0:3] = 0.5
        nodes[[0, 1, 2], [0, 1, 2]] = 0.0
        nodes[3:6] = 0.0
        nodes[[3, 4, 5], [0, 1, 2]] = 0.5
        for row, b in ((6, 0.1005267652252045), (10, 0.3143728734931922)):
            a = {0.1005267652252045: 0.6984197043243866, 0.3143728734931922: 0.0568813795204234}[b]
            nodes[row:row + 4] = b
            nodes[[row, row + 2, row + 3], [0, 2, 1]] = a
        weights = np.repeat([0.0190476190476190, 0.0885898247429807, 0.1328387466855907], [6, 4, 4]) / 6.0
        return nodes, weights
    raise NotImplementedError("rules of order 2, 3 and 4 (what the reference provides, Qudrature.py:6-45)")
