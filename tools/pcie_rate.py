#!/usr/bin/env python3
"""Diagnostic: throughput of the drop-in compatibility path, where the caller owns the state on the HOST and every call
crosses PCIe: per step ``LocalK.dot(d0)`` (upload d0, download f) + the update on caller arrays
(upload f, d0, dn; download d1) - what ``Tools.Dynamic_solver.parallel_explicit_solver_dis_pre`` does per call - against
the device-resident ``saa_step``.  Also times the HIP set-up kernels against the NumPy closed forms."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd import fem_setup as fs  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
mesh = structured_beam(n)
t0 = time.perf_counter()
dev = fs.device_setup_fields(mesh.points, mesh.tets, 1.0, 0.5)
t1 = time.perf_counter()
dev = fs.device_setup_fields(mesh.points, mesh.tets, 1.0, 0.5)
t2 = time.perf_counter()
host = fs.host_setup_fields(mesh.points, mesh.tets, 1.0, 0.5)
t3 = time.perf_counter()
print(f"set-up fields, {len(mesh.tets)} tets: HIP kernels {t2 - t1:.3f} s (first call {t1 - t0:.3f} s, host arrays in and out) "
      f"against NumPy closed forms {t3 - t2:.3f} s; equal to {np.abs(dev[0] - host[0]).max() / np.abs(host[0]).max():.1e}")
sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0)
ne = len(mesh.tets)
d0 = np.zeros((sol.n_dof, 1))
dn = np.zeros_like(d0)
tn, steps = 0.0, 20
for _ in range(3):
    sol.cd_update(sol.internal_force(d0), d0, dn, tn)
t0 = time.perf_counter()
for _ in range(steps):
    d1 = sol.cd_update(sol.internal_force(d0), d0, dn, tn)
    dn, d0, tn = d0, d1, tn + dt
el = time.perf_counter() - t0
print(f"host-owned state (PCIe every call): {el / steps * 1e3:.2f} ms/step = {ne * steps / el:.3e} element-updates/s")
sol.step(2000)
ms = sol.time_steps(5000)
print(f"device-resident saa_step           : {ms / 5000 * 1e3:.2f} us/step = {ne * 5000 / (ms * 1e-3):.3e} element-updates/s")
