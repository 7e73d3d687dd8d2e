#!/usr/bin/env python3
"""Diagnostic: how much delivery time the peer exchange of the resident kernel tolerates.  The middle slab of the 8-GPU
partition steps through the exchange with loop-back neighbours; a diagnostic build holds every collected value back until
`latency` has passed since the push (-DSAA_PEER_EMULATE_LATENCY=<ticks of 10 ns>), a stand-in for xGMI's delivery time.

    python tools/peer_latency.py [latency in us, default 0]      (one latency per process: each needs its own build)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library  # noqa: E402

lat_us = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
use_diag_library([f"-DSAA_PEER_EMULATE_LATENCY={int(round(lat_us * 100))}"] if lat_us > 0 else ["-DSAA_PEER_LATENCY_BASE"])
import torch  # noqa: E402

from bench import N_FOR_GPUS, build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

sol = build_rank_solver(structured_beam(N_FOR_GPUS[8]), 8, 3, 0)[0]
sol.peer_attach_loopback(2)


def timed(fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(steps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


sol.step_peer(20000)
sol.step(20000)
pe, pl = [], []
for _ in range(5):
    pe.append(timed(sol.step_peer, 3000))
    pl.append(timed(sol.step, 3000))
print(f"emulated delivery time {lat_us:.1f} us: exchange-free step {sorted(pl)[2]:.2f} us, step through the peer exchange "
      f"{sorted(pe)[2]:.2f} us")
