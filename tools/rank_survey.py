#!/usr/bin/env python3
"""Diagnostic: every rank's partition of the N = 2, 4, 8 bench configurations on ONE GPU - plan, whether the
resident kernel holds it, its exchange-free step time and its step time through the peer exchange with loop-back
neighbours (the slowest rank paces a synchronised run).

    python tools/rank_survey.py [N ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import N_FOR_GPUS, build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402


def timed(fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(steps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


for world in ([int(a) for a in sys.argv[1:]] or (2, 4, 8)):
    mesh = structured_beam(N_FOR_GPUS[world])
    for r in range(world):
        sol, lay, _, _ = build_rank_solver(mesh, world, r, 0)
        st, ri = sol.plan_stats(), sol.resident_kernel_info()
        # clocks settle over the first tens of milliseconds of load: warm both paths, then alternate them
        sol.peer_attach_loopback(2)
        sol.step_peer(20000)
        sol.step(20000)
        pe, pl = [], []
        for _ in range(5):
            pe.append(timed(sol.step_peer, 3000))
            pl.append(timed(sol.step, 3000))
        plain, peer = sorted(pl)[2], sorted(pe)[2]
        sol.synchronize()
        print(f"N={world} rank {r}: tets {len(lay.cells_local)} nodes {len(lay.nodes)} shared {len(lay.shared_local)} "
              f"blocks {st['n_blocks']} max_owned {st['max_owned']} max_local {st['max_local']} conflict "
              f"{st['lds_conflict_factor']:.3f} resident {ri['capable']} lds {ri['lds_bytes']}  step {plain:.2f} us, "
              f"with peer exchange (loop-back) {peer:.2f} us", flush=True)
        sol.close()
