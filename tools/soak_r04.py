#!/usr/bin/env python3
"""Soak of the two schedules round 4 added (long runs, looking for a hang, a drift or a wrong dependency):
  * split stepping of the 8.2M-tet beam (three streams, events) against one launch per step: 100 000 steps each;
  * the Delaunay beam (joint numbering, augmenting-path pairing) through the resident kernel against the one-launch-per-step
    kernel: 500 000 steps each.                                                     python tools/soak_r04.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam, structured_beam  # noqa: E402


def run(sol, steps, chunk, setter):
    zero = np.zeros(sol.n_dof)
    out = []
    for mode in (1, 0):
        setter(sol, mode)
        sol.set_state(zero, zero, 0.0)
        t0 = time.time()
        for _ in range(steps // chunk):
            sol.step(chunk)
        d0, _, tn = sol.get_state()
        out.append((d0, tn, time.time() - t0))
    (a, ta, sa), (b, tb, sb) = out
    assert np.isfinite(a).all() and ta == tb
    return np.linalg.norm(a - b) / np.linalg.norm(b), np.abs(b).max(), sa, sb


sol = build_rank_solver(structured_beam(38), 1, 0, 0)[0]
err, mx, s1, s0 = run(sol, 100000, 5000, lambda s, m: s.set_option("split_stepping", m))
print(f"8.2M tets, 100 000 steps from rest: split stepping {s1:.1f} s, one launch per step {s0:.1f} s; rel-L2 between the end states "
      f"{err:.3e}, max|d| {mx:.3e}", flush=True)
sol.close()
sol = build_rank_solver(delaunay_beam(19), 1, 0, 0)[0]
err, mx, s1, s0 = run(sol, 500000, 25000, lambda s, m: s.set_resident_kernel(bool(m)))
print(f"Delaunay beam, 500 000 steps from rest: resident kernel {s1:.1f} s, one launch per step {s0:.1f} s; rel-L2 between the end "
      f"states {err:.3e}, max|d| {mx:.3e}", flush=True)
sol.close()
