"""Not a pytest file: the 8 230 800-tet beam on ONE GPU (the `cache_exceeding` leg of bench.py: fused kernel, 2048 blocks,
split stepping on three streams) against the CPU oracle - the reference's element matrices applied element by element
(fem_oracle.MatrixFreeStiffness, 9.5 GB of them) - from a rough state, which the test suite leaves to size-independent
properties because an oracle step takes a second here.      python tests/long_parity_8M.py [steps]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fem_oracle as fo  # noqa: E402
from bench import ALPHA, build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd import fem_setup as fs  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mesh = structured_beam(38)
sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0)
lmd, mu = fs.lame(1e6, 0.3)
t0 = time.time()
K = fo.MatrixFreeStiffness(lay.cells_local, mesh.points[lay.nodes], lmd, mu, threads=16)
l_M, F, _ = fs.rank_fields(mesh.points, mesh.tets, lay, 1.0, 0.5)
print(f"{len(mesh.tets)} tets, plan {sol.plan_stats()}, resident {sol.resident_kernel_info()['capable']}; oracle operator in "
      f"{time.time() - t0:.0f} s", flush=True)
rng = np.random.default_rng(38)
d0 = rng.uniform(-1e-4, 1e-4, size=(sol.n_dof, 1))
dn = d0 + rng.uniform(-1e-6, 1e-6, size=(sol.n_dof, 1))
d0[lay.dirichlet_dofs] = 0
dn[lay.dirichlet_dofs] = 0
f = sol.internal_force(d0)
print(f"K.d: rel-L2 GPU vs oracle = {np.linalg.norm(f - K.dot(d0)) / np.linalg.norm(f):.3e}", flush=True)
marks = [m for m in (10, 30, 60, 100, 200) if m <= steps]
tn, o0, on, done = 0.25, d0, dn, 0
got = {}
for split in (1, 0):
    sol.set_option("split_stepping", split)
    sol.set_state(d0, dn, 0.25)
    for m in marks:
        sol.step(m - (marks[marks.index(m) - 1] if marks.index(m) else 0))
        got[(split, m)] = sol.get_state()[0]
t0 = time.time()
for m in marks:
    for _ in range(m - done):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        on, o0, tn = o0, o1, tn + dt
    done = m
    a, b = got[(1, m)], got[(0, m)]
    print(f"step {m:4d}: rel-L2 vs oracle: split stepping {np.linalg.norm(a - o0) / np.linalg.norm(o0):.3e}, one launch per step "
          f"{np.linalg.norm(b - o0) / np.linalg.norm(o0):.3e}; split vs plain {np.linalg.norm(a - b) / np.linalg.norm(b):.3e}   "
          f"({time.time() - t0:.0f} s of oracle steps)", flush=True)
