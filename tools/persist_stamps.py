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
from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

# usage: persist_stamps.py [n [parts rank]]   (parts > 1: the x-slab partition `rank` of `parts`, exchange-free steps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
steps = 1000
mesh = structured_beam(n)
sol, lay, _, _ = build_rank_solver(mesh, parts, prank, 0)
st = sol.plan_stats()
waves = st["n_blocks"] * st["threads"] // 64
dbg = torch.zeros(8 * waves, dtype=torch.float64, device="cuda")  # reinterpreted as uint64 by the kernel
import ctypes as C
from synchronization_avoiding_algorithms_amd import _lib
lib = _lib.load()
lib.saa_debug_set_stamp_buffer.restype = C.c_int
sol.step(200)
sol.synchronize()
_lib.check(lib.saa_debug_set_stamp_buffer(sol._h, C.c_void_p(dbg.data_ptr())))
sol.step(steps)   # one launch of `steps` steps: the counts are per launch
sol.synchronize()
_lib.check(lib.saa_debug_set_stamp_buffer(sol._h, None))
t = dbg.cpu().numpy().view(np.uint64).reshape(waves, 8).astype(np.float64) / steps
names = ["round 1 (interior items, halo fetch issued)", "-", "settle halo -> LDS", "barrier (halo)",
         "other interior + boundary items", "barrier (slowest wave)", "update", "barrier (end of step)"]
tot = t.sum(axis=1)
print(f"n={n} plan {st}")
print(f"shader-clock cycles per step per wave, median total {np.median(tot):.0f}")
for j, nm in enumerate(names):
    print(f"  {nm:34s} median {np.median(t[:, j]):8.0f}  p90 {np.percentile(t[:, j], 90):8.0f}  max {t[:, j].max():8.0f}")
ms = sol.time_steps(1000)
print(f"stamped build: {ms:.3f} us/step")
