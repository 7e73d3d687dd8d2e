#!/usr/bin/env python3
"""Diagnostic: where a wave of the fused step kernel spends its cycles (in-kernel s_memtime stamps, ABLATE 8).
Shares, not absolute times: the stamps drain the LDS queue and forbid overlaps the real kernel has."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library  # noqa: E402

use_diag_library()  # diagnostic build of the library; the product .so has none of the saa_debug_* entry points
import numpy as np  # noqa: E402

from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd import _lib  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
mesh = structured_beam(n)
sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0)
rng = np.random.default_rng(0)
d = rng.uniform(-1e-4, 1e-4, size=sol.n_dof)
sol.set_state(d, d, 0.5)
lib = _lib.load()
lib.saa_debug_time_ablated.restype = C.c_int
lib.saa_debug_time_ablated.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
lib.saa_debug_read_stamps.restype = C.c_int
lib.saa_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
st = sol.plan_stats()
ms = C.c_double()
lib.saa_debug_time_ablated(sol._h, 8, 20, C.byref(ms))
print(f"stamped build: {ms.value / 20 * 1e3:.2f} us/launch (real kernel is faster)")
waves = st["n_blocks"] * st["threads"] // 64
buf = np.zeros(12 * waves, dtype=np.uint64)
assert lib.saa_debug_read_stamps(sol._h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
T = buf.reshape(waves, 12).astype(np.float64)
names = ["item: 12 reads (issue+arrive)", "item: VALU tet A", "item: flush a + read b", "item: VALU tet B",
         "item: 12 atomics (issue+drain)", "-", "kernel: staging + barrier", "kernel: interior loop",
         "kernel: halo to LDS + barrier", "kernel: boundary loop", "kernel: wait for slowest wave", "kernel: update"]
tot = T[:, 6:].sum(axis=1)
print(f"cycles per wave (median over {waves} waves); total {np.median(tot):.0f}")
for j, nm in enumerate(names):
    if nm != "-":
        print(f"  {nm:34s} median {np.median(T[:, j]):9.0f}   p90 {np.percentile(T[:, j], 90):9.0f}   max {T[:, j].max():9.0f}")
items = (T[:, 7] + T[:, 9])
print("  item-phase sum of parts / loops:", np.median(T[:, :5].sum(axis=1)), "/", np.median(items))
