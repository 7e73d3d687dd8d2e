#!/usr/bin/env python3
"""Diagnostic: step time over (block_nodes, threads) on the n-refined beam: calls of 1000 steps (resident kernel when
the plan admits it, else the fused kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
mesh = structured_beam(n)
rng = np.random.default_rng(0)
for bn in (int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "744,372,248,186".split(","))):
    for th in (int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "256,512,1024".split(","))):
        try:
            sol, lay, _, dt = build_rank_solver(mesh, 1, 0, 0, bn, th)
        except Exception as exc:  # noqa: BLE001
            print(bn, th, "failed:", exc)
            continue
        d = rng.uniform(-1e-4, 1e-4, size=sol.n_dof)
        sol.set_state(d, d, 0.5)
        sol.time_steps(200)
        ms = sol.time_steps(1000)
        st = sol.plan_stats()
        info = sol.resident_kernel_info()
        print(f"block_nodes {bn:5d} threads {th:5d}: {ms:8.3f} us/step  blocks {st['n_blocks']} owned {st['max_owned']} "
              f"local {st['max_local']} copies {st['n_elem_copies']} lds {st['lds_bytes']} conflict "
              f"{st['lds_conflict_factor']:.3f} resident {info['capable']} ({info['lds_bytes']} B)", flush=True)
        sol.close()
