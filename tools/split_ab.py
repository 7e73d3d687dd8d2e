#!/usr/bin/env python3
"""Diagnostic: the 8.2M-tet beam on one GPU (fused kernel, 2048 blocks on 512 slots) stepped with one launch of all blocks
per step against split stepping (saa_api.cpp: three block sets on three streams, saa_set_option("split_stepping")),
alternating on one box; and the two end states against each other.      python tools/split_ab.py [n] [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "SAA_SPLIT_LEFT_PCT" in os.environ:  # (the share of the left set is a switch of the diagnostic build only)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _diag import use_diag_library

    use_diag_library()
from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 38
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mesh = structured_beam(n)
sol = build_rank_solver(mesh, 1, 0, 0)[0]
print("left share", os.environ.get("SAA_SPLIT_LEFT_PCT", "45 (default)"), "plan", sol.plan_stats(), "resident", sol.resident_kernel_info()["capable"], flush=True)
rng = np.random.default_rng(0)
d = rng.uniform(-1e-5, 1e-5, size=sol.n_dof)
ends = {}
for mode in (0, 1):
    sol.set_option("split_stepping", mode)
    sol.set_state(d, d, 0.5)
    sol.step(300)
    ends[mode] = sol.get_state()[0]
print(f"300 steps, split against plain: rel-L2 {np.linalg.norm(ends[1] - ends[0]) / np.linalg.norm(ends[0]):.3e}", flush=True)
sol.time_steps(500)
for r in range(rounds):
    for mode in (0, 1):
        sol.set_option("split_stepping", mode)
        sol.time_steps(200)
        us = [1e3 * sol.time_steps(4000) / 4000 for _ in range(2)]
        print(f"{'split stepping (3 streams)' if mode else 'one launch per step      '}: " + " ".join(f"{u:.2f}" for u in us) + " us/step", flush=True)
