#!/usr/bin/env python3
"""Robustness sweep of the plan paths round 4 added (pairing by augmenting paths, numbering decided while packing): random
Delaunay meshes of several sizes and block sizes - K.d and 25 steps of both kernels against the oracle's assembled matrix.

    python tools/fuzz_unstructured.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import fem_oracle as fo  # noqa: E402
from test_gpu_parity import _delaunay_mesh, _serial_solver  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(2026)
worst = 0.0
for c in range(cases):
    n_points = int(rng.integers(800, 9000))
    block_nodes = int(rng.choice([0, 48, 96, 200, 400]))
    threads = int(rng.choice([0, 128, 256, 512]))
    mesh = _delaunay_mesh(n_points, seed=1000 + c)
    sol, lay, dt, lumped, fpre = _serial_solver(mesh, block_nodes=block_nodes, threads=threads)
    st = sol.plan_stats()
    ranks, odt, _, _ = fo.setup_problem(mesh.points, mesh.tets, mesh.triangles, 1, np.zeros(len(mesh.tets), dtype=int))
    rp = ranks[0]
    d = rng.uniform(-1e-3, 1e-3, size=(sol.n_dof, 1))
    e_k = np.linalg.norm(sol.internal_force(d) - rp.K.dot(d)) / np.linalg.norm(rp.K.dot(d))
    sol.set_loads(rp.F, rp.l_M)
    d0 = rng.uniform(-1e-6, 1e-6, size=(sol.n_dof, 1))
    d0[rp.dirichlet] = 0
    tn, o0, on = 0.3, d0, d0
    for _ in range(25):
        o1 = fo.explicit_step(rp.K, rp.F, rp.dirichlet, tn, dt, o0, on, rp.l_M, 0.5)
        on, o0, tn = o0, o1, tn + dt
    errs = []
    for resident in (True, False):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, d0, 0.3)
        sol.step(25)
        errs.append(np.linalg.norm(sol.get_state()[0] - o0) / np.linalg.norm(o0))
    worst = max(worst, e_k, *errs)
    print(f"case {c:2d}: {len(mesh.points):5d} nodes {len(mesh.tets):6d} tets block_nodes {block_nodes:3d} threads {threads:3d} -> blocks "
          f"{st['n_blocks']:4d} items {st['n_items']:6d} pairs {st['n_pairs']:6d} renumbered {st['n_renumbered']:4d} conflict "
          f"{st['lds_conflict_factor']:.2f}/{st['lds_atomic_conflict_factor']:.2f}  K.d {e_k:.1e}  steps resident {errs[0]:.1e} fused {errs[1]:.1e}",
          flush=True)
    sol.close()
print(f"worst relative error over {cases} cases: {worst:.2e}")
assert worst < 1e-10
