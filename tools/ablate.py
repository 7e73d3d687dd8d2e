#!/usr/bin/env python3
"""Diagnostic: time the fused step kernel with one phase removed (see ABLATE in saa_kernels.hip)."""
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
bn = int(sys.argv[2]) if len(sys.argv) > 2 else 0
th = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mesh = structured_beam(n)
sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0, bn, th)
rng = np.random.default_rng(0)
d = rng.uniform(-1e-4, 1e-4, size=sol.n_dof)
sol.set_state(d, d, 0.5)
lib = _lib.load()
lib.saa_debug_time_ablated.restype = C.c_int
lib.saa_debug_time_ablated.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
names = {0: "full", 1: "no LDS atomics", 2: "no indexed LDS reads", 3: "no staging loads",
         4: "no update phase", 5: "no element phase", 6: "element VALU only", 7: "element LDS only",
         9: "8 steps of blocks, 1 launch", 10: "element phase only (3 + 4)"}
print("plan", sol.plan_stats())
ms = C.c_double()
for v in (0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 0):
    lib.saa_debug_time_ablated(sol._h, v, 200, C.byref(ms))
    lib.saa_debug_time_ablated(sol._h, v, 1000, C.byref(ms))
    print(f"variant {v:2d} ({names[v]:26s}): {ms.value:8.3f} us/launch")
# Plans with two workgroups per CU (the 8.2M-tet beam on one GPU): the same variants with room for ONE workgroup per CU
# only (unused dynamic LDS on top of the image) - what the element phase delivers at 8 waves per CU instead of 16, i.e.
# what a workgroup whose waves were split into a staging / updating half and a computing half would have to live on.
st = sol.plan_stats()
if 2 * st["lds_bytes"] <= 160 * 1024 and st["n_blocks"] > 256:
    os.environ["SAA_ABLATE_EXTRA_LDS"] = str(160 * 1024 // 2 - st["lds_bytes"] + 2048)
    for v in (0, 5, 10, 6, 7, 0):
        lib.saa_debug_time_ablated(sol._h, v, 200, C.byref(ms))
        lib.saa_debug_time_ablated(sol._h, v, 1000, C.byref(ms))
        print(f"ONE workgroup per CU, variant {v:2d} ({names[v]:26s}): {ms.value:8.3f} us/launch")
    os.environ.pop("SAA_ABLATE_EXTRA_LDS")
