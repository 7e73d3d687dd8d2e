#!/usr/bin/env python3
"""Diagnostic: where a step of the resident kernel spends its cycles (diagnostic build with -DSAA_PERSIST_STAMPS,
made on the fly: libsaa_hip_diag.so)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library  # noqa: E402

use_diag_library(["-DSAA_PERSIST_STAMPS"])  # diagnostic build of the library; the product .so has none of the saa_debug_* entry points
from bench import bench_mesh, build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

# usage: persist_stamps.py [n [parts rank [peer]]]   (parts > 1: the x-slab partition `rank` of `parts`; exchange-free steps,
#                                                      or - "peer" - steps through the peer exchange with loop-back)
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(pos[0]) if len(pos) > 0 else 19
parts = int(pos[1]) if len(pos) > 1 else 1
prank = int(pos[2]) if len(pos) > 2 else 0
peer = len(pos) > 3 and pos[3] == "peer"
steps = 1000
mesh = bench_mesh(n, "jittered") if "--mesh=jittered" in sys.argv else (bench_mesh(n, "delaunay") if "--mesh=delaunay" in sys.argv else structured_beam(n))
sol, lay, _, _ = build_rank_solver(mesh, parts, prank, 0)
st = sol.plan_stats()
waves = st["n_blocks"] * st["threads"] // 64
# reinterpreted as uint64 by the kernel; in peer mode the same buffer first receives the history rows of the shared dofs
dbg = torch.zeros(max(8 * waves, steps * 3 * len(lay.shared_local) if peer else 0), dtype=torch.float64, device="cuda")
step = sol.step
if peer:
    sol.peer_attach_loopback(2)
    step = sol.step_peer
import ctypes as C
from synchronization_avoiding_algorithms_amd import _lib
lib = _lib.load()
lib.saa_debug_set_stamp_buffer.restype = C.c_int
step(200)
sol.synchronize()
_lib.check(lib.saa_debug_set_stamp_buffer(sol._h, C.c_void_p(dbg.data_ptr())))
step(steps)   # one launch of `steps` steps: the counts are per launch
sol.synchronize()
_lib.check(lib.saa_debug_set_stamp_buffer(sol._h, None))
t = dbg[:8 * waves].cpu().numpy().view(np.uint64).reshape(waves, 8).astype(np.float64) / steps
names = ["round 1 (interior items, halo fetch issued)", "-", "settle halo -> LDS", "barrier (halo)",
         "other interior + boundary items", "barrier (slowest wave)", "update", "barrier (end of step)"]
tot = t.sum(axis=1)
print(f"n={n} plan {st}")
print(f"shader-clock cycles per step per wave, median total {np.median(tot):.0f}")
for j, nm in enumerate(names):
    print(f"  {nm:34s} median {np.median(t[:, j]):8.0f}  p90 {np.percentile(t[:, j], 90):8.0f}  max {t[:, j].max():8.0f}")
# per workgroup (16 waves each): which blocks are slow in the update phase (shared nodes sit in the face blocks)
wg = t.reshape(st["n_blocks"], -1, 8).mean(axis=1)
order = np.argsort(wg[:, 6])
print("  update phase per workgroup: min %.0f median %.0f max %.0f;  items phases (0+4): min %.0f median %.0f max %.0f" % (
    wg[:, 6].min(), np.median(wg[:, 6]), wg[:, 6].max(), (wg[:, 0] + wg[:, 4]).min(), np.median(wg[:, 0] + wg[:, 4]),
    (wg[:, 0] + wg[:, 4]).max()))
print("  slowest update workgroups:", [(int(b), int(wg[b, 6]), int(wg[b, 0] + wg[b, 4])) for b in order[-5:]])
if not peer:
    ms = sol.time_steps(1000)
    print(f"stamped build: {ms:.3f} us/step")
for a in sys.argv[1:]:
    if a.startswith("--json="):  # medians for tools/onchip_summary.py
        import json

        med = [float(np.median(t[:, j])) for j in range(8)]
        with open(a.split("=", 1)[1], "w") as fh:
            json.dump({"n": n, "cycles_per_step_median_total": float(np.median(tot)), "phases": dict(zip(names, med)),
                       "barrier_cycles": med[3] + med[5] + med[7]}, fh, indent=1)
