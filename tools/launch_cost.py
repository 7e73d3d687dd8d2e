#!/usr/bin/env python3
"""Diagnostic: the fixed cost of one resident launch.  Calls of k steps (one launch each), queued back to back, wall
clock per call for several k; least-squares fit  t(k) = a + b*k  - a is what a launch costs beyond its steps (dispatch,
image staging, first halo round, state written back), b the step itself.  The driver's command (--steps 20) pays a/20
per step on top of b.

    python tools/launch_cost.py [n] [--mesh=jittered]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
kind = "jittered" if "--mesh=jittered" in sys.argv else "structured"
sys.argv = ["bench.py"]
import numpy as np  # noqa: E402

from bench import bench_mesh, build_rank_solver  # noqa: E402

n = int(args[0]) if args else 19
sol = build_rank_solver(bench_mesh(n, kind), 1, 0, 0)[0]
assert sol.resident_kernel_info()["capable"]
rng = np.random.default_rng(0)
d = rng.uniform(-1e-5, 1e-5, size=sol.n_dof)
sol.set_state(d, d, 0.5)
sol.step(2000)
ks = [8, 10, 14, 20, 28, 40, 60, 100, 200, 400]
rows = []
for rep in range(3):
    for k in ks:
        calls = max(20, 40000 // k)
        sol.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            sol.step(k)
        sol.synchronize()
        rows.append((k, 1e6 * (time.perf_counter() - t0) / calls))
A = np.array([[1.0, k] for k, _ in rows])
y = np.array([t for _, t in rows])
(a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
for k in ks:
    ts = [t for kk, t in rows if kk == k]
    print(f"k = {k:4d}: {min(ts):9.2f} us per call (min of 3), {min(ts) / k:7.3f} us/step, fit {a + b * k:9.2f}")
print(f"fit: t(k) = {a:.2f} us + {b:.4f} us * k   ({kind} mesh, n = {n})")
