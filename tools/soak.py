#!/usr/bin/env python3
"""Soak of the resident kernel's stamped-halo protocol at the bench's size: the 1 028 850-tet beam stepped N times through
1000-step launches of the resident kernel and through one launch of the fused kernel per step, from the same state; then
the middle slab of the 8-GPU partition through the peer exchange (loop-back) in both kernels.

    python tools/soak.py [steps=1000000]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def pair(mesh, world, rank):
    res = build_rank_solver(mesh, world, rank, 0)[0]
    fus = build_rank_solver(mesh, world, rank, 0)[0]
    fus.set_resident_kernel(False)
    assert res.resident_kernel_info()["capable"] and not fus.resident_kernel_info()["capable"]
    return res, fus


mesh = structured_beam(19)
res, fus = pair(mesh, 1, 0)
for sol in (res, fus):
    t0 = time.time()
    sol.step(steps)
    sol.synchronize()
    print(f"  {'resident' if sol is res else 'fused   '}: {steps} steps in {time.time() - t0:.1f} s", flush=True)
a, b = res.get_state()[0], fus.get_state()[0]
print(f"n=19 ({len(mesh.tets)} tets), from rest under the ramped load, {steps} steps: rel-L2(resident vs fused) = {rel(a, b):.3e}, "
      f"max|d| = {np.abs(b).max():.4g}, time {res.get_state()[2]!r} == {fus.get_state()[2]!r}")
res.close()
fus.close()

mesh = structured_beam(38)
res, fus = pair(mesh, 8, 3)
n = steps // 4
for sol in (res, fus):
    sol.peer_attach_loopback(2)
    t0 = time.time()
    sol.step_peer(n)
    sol.synchronize()
    print(f"  {'resident' if sol is res else 'fused   '} PEER: {n} steps in {time.time() - t0:.1f} s", flush=True)
a, b = res.get_state()[0], fus.get_state()[0]
print(f"rank 3 of 8 of n=38, peer exchange with one loop-back neighbour, {n} steps: rel-L2(resident vs fused) = {rel(a, b):.3e}, "
      f"max|d| = {np.abs(b).max():.4g}")
