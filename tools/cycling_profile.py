#!/usr/bin/env python3
"""Diagnostic: where the tasks of the cycling kernel spend their time (four wall-clock stamps per (step, block))."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library  # noqa: E402

use_diag_library(tuple(a for a in sys.argv[3:] if a.startswith("-D")))
import numpy as np  # noqa: E402

from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd import _lib  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 38
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mesh = structured_beam(n)
sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0, 0, 0)
rng = np.random.default_rng(0)
d = rng.uniform(-1e-4, 1e-4, size=sol.n_dof)
sol.set_state(d, d, 0.5)
st = sol.plan_stats()
print("plan", st, sol.multistep_kernel_info())
nb = st["n_blocks"]
lib = _lib.load()
lib.saa_debug_cycling_profile.restype = C.c_int
lib.saa_debug_cycling_profile.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
buf = np.zeros((steps, nb, 4), dtype=np.int64)
sol.step(steps)
_lib.check(lib.saa_debug_cycling_profile(sol._h, steps, buf.ctypes.data, buf.size))
t = (buf - buf.min()) * 1e-2  # us
print(f"launch: {t.max():.1f} us for {steps} steps = {t.max() / steps:.2f} us/step")
dur = t[..., 3] - t[..., 0]
own = t[..., 1] - t[..., 0]
halo = t[..., 2] - t[..., 1]
rest = t[..., 3] - t[..., 2]
mid = slice(steps // 4, steps)  # steady state
for name, v in (("task total", dur), ("start -> own step seen (coordinates staged, own flag)", own),
                ("-> halo records in LDS (displacements, halo waits, interior items)", halo),
                ("-> flag published (boundary items, update, drain)", rest)):
    q = v[mid]
    print(f"  {name:75s} mean {q.mean():7.2f}  median {np.median(q):7.2f}  p90 {np.percentile(q, 90):7.2f}  max {q.max():8.2f} us")
# step skew: when does each block start step s, relative to the mean start of that step
start = t[..., 0]
skew = start[mid] - start[mid].mean(axis=1, keepdims=True)
print(f"  start skew inside a step: std {skew.std():.2f} us, span {np.percentile(skew, 1):.1f} .. {np.percentile(skew, 99):.1f} us")
per_xcd = start[mid].reshape(start[mid].shape[0], 8, -1).mean(axis=2)
print("  mean start of a step per XCD run, relative:", " ".join(f"{v:7.2f}" for v in (per_xcd - per_xcd.mean(axis=1, keepdims=True)).mean(axis=0)))
slow = np.argsort(dur[mid].mean(axis=0))[-8:]
print("  slowest blocks (mean task us):", [(int(b), round(float(dur[mid][:, b].mean()), 1), round(float(own[mid][:, b].mean()), 1),
                                           round(float(halo[mid][:, b].mean()), 1)) for b in slow])
