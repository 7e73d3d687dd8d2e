"""Counterpart of the reference's ``Tools/commons.py`` (helpers of the explicit path only)."""
from __future__ import annotations

import numpy as np

from ..fem_setup import meshsize as _meshsize
from ..fem_setup import node_to_dof as _node_to_dof


def linear_ramp(t):
    """Load ramp ending at t = 1 s (``commons.py:7-11``)."""
    return t if t <= 1 else 1.0


class elasticity:
    """Problem parameters (``commons.py:15-41``): Lame constants, density, body force, ramp flag."""

    def __init__(self, lmd, mu, rho, fz, R):
        self.lmd, self.mu, self.rho, self.fz, self.R = lmd, mu, rho, fz, R

    def D(self):
        d = np.zeros((6, 6))
        d[:3, :3] = self.lmd
        d[range(3), range(3)] += 2.0 * self.mu
        d[range(3, 6), range(3, 6)] = self.mu
        return d

    def f(self, X, t):
        s = linear_ramp(t) if self.R else 1.0
        return np.array([[0.0], [-self.fz * s], [-self.fz * s]])


class Time_integration_displacement:
    """``commons.py:47-55``: ``d0 = d^n``, ``dn = d^(n-1)``."""

    def __init__(self, tn, dt, d0, dn):
        self.tn, self.dt, self.d0, self.dn = tn, dt, d0, dn

    def tn_plus_1(self):
        return self.tn + self.dt


def node_to_dof(d, ls, P):
    """``commons.py:66-71`` (list of ints, like the reference)."""
    P = np.asarray(P, dtype=np.int64).reshape(-1)
    if d == 3 and list(ls) == [0, 1, 2]:
        return _node_to_dof(P).tolist()
    return (d * P[:, None] + np.asarray(ls, dtype=np.int64)[None, :]).ravel().tolist()


def Meshsize(Element, Points):
    """``commons.py:79-90``: 2*min_edge/sqrt(24)."""
    return _meshsize(np.asarray(Points, dtype=np.float64), np.asarray(Element, dtype=np.int64))


def lumping_to_vec(M):
    """Row sums as a ``(n,1)`` vector (``commons.py:103-107``); accepts dense or scipy-sparse ``M``."""
    return np.asarray(M.sum(axis=1), dtype=np.float64).reshape(-1, 1)
