"""Not a pytest file: a longer full-size parity run for profiles/ (python tests/long_parity_fullsize.py [steps] [delaunay]).

The 1 028 850-tet beam of BASELINE.json configs[2] from rest under the ramped load, resident kernel (1000-step
launches) against the CPU oracle (reference element matrices applied element by element, fem_oracle.MatrixFreeStiffness):
rel-L2 of the displacement field at a few step counts."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fem_oracle as fo  # noqa: E402
from bench import ALPHA, build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam, structured_beam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
marks = [m for m in (100, 300, 1000, 2000, 3000, 5000, 10000) if m <= steps]
# "delaunay": the same box meshed without any lattice (bench.py --mesh delaunay, the `unstructured` leg)
mesh = delaunay_beam(19) if "delaunay" in sys.argv[2:] else structured_beam(19)
sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0)
from synchronization_avoiding_algorithms_amd import fem_setup as fs  # noqa: E402

lmd, mu = fs.lame(1e6, 0.3)
K = fo.MatrixFreeStiffness(lay.cells_local, mesh.points[lay.nodes], lmd, mu)
l_M, F, _ = fs.rank_fields(mesh.points, mesh.tets, lay, 1.0, 0.5)
print(f"{len(mesh.tets)} tets, dt = {dt!r}, plan {sol.plan_stats()}, resident {sol.resident_kernel_info()}", flush=True)
o0 = np.zeros((sol.n_dof, 1))
on = np.zeros_like(o0)
tn, done, t0 = 0, 0, time.time()
for m in marks:
    for _ in range(m - done):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        on, o0, tn = o0, o1, tn + dt
    sol.step(m - done)
    done = m
    g0 = sol.get_state()[0]
    print(f"step {m:6d}: rel-L2 GPU vs oracle = {np.linalg.norm(g0 - o0) / np.linalg.norm(o0):.3e}   max|d| = "
          f"{np.abs(o0).max():.3e}   ({time.time() - t0:.0f} s)", flush=True)
