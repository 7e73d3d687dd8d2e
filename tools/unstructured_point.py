#!/usr/bin/env python3
"""Diagnostic: what an unstructured mesh of the bench's size costs - the 1 028 850-tet beam with jittered nodes and
shuffled node / element numbering (bench.py --mesh jittered uses the same kind of mesh).  SAA_PLAN_SHAPE_PAIRS=0
SAA_PLAN_LATTICE_ORDERS=0 switch the plan back to round 2's behaviour (elements paired in list order, block nodes numbered
by exact coordinates: no pattern classes on such a mesh, the greedy stage does all the work)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

if any(k.startswith("SAA_PLAN_") for k in os.environ):  # those switches exist in the diagnostic build only
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _diag import use_diag_library

    use_diag_library()

from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402
from test_gpu_parity import _scrambled_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
for name, mesh in (("structured", structured_beam(n)), ("jittered + shuffled", _scrambled_mesh(n, 0)[0])):
    t0 = time.time()
    sol = build_rank_solver(mesh, 1, 0, 0)[0]
    t1 = time.time()
    rng = np.random.default_rng(0)
    d = rng.uniform(-1e-5, 1e-5, size=sol.n_dof)
    sol.set_state(d, d, 0.5)
    sol.time_steps(1000)
    us = np.mean([sol.time_steps(1000) for _ in range(5)])
    st = sol.plan_stats()
    print(f"{name:20s}: set-up {t1 - t0:.1f} s, {us:.2f} us/step, conflict factor reads {st['lds_conflict_factor']:.3f} atomics "
          f"{st['lds_atomic_conflict_factor']:.3f}, element copies {st['n_elem_copies'] / len(mesh.tets):.3f}x in {st['n_items']} "
          f"items ({st['n_pairs']} pairs, {st['n_by_construction']} in clash-free halves by construction), blocks "
          f"{st['n_blocks']}, resident {sol.resident_kernel_info()['capable']}", flush=True)
    sol.close()
