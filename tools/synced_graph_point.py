#!/usr/bin/env python3
"""Diagnostic: what the launch route of saa_step_synced costs - the middle slab of the 8-GPU partition of the 8.2M-tet beam
(1 031 016 tets, 3042 shared nodes, all-reduce buffer of 31 941 doubles) stepped through fused kernel -> ncclAllReduce ->
finish kernel with a ONE-rank RCCL communicator (so the collective itself is as cheap as it gets and what is left is the
cost of launching three things per step): eager launches against replayed HIP graphs of three steps each."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

mesh = structured_beam(38)
sol, lay, gshared, _ = build_rank_solver(mesh, 8, 3, 0)
iface = torch.zeros(3 * len(gshared), dtype=torch.float64, device="cuda")
sol.set_interface_buffer(iface)
sol.set_stream(torch.cuda.current_stream().cuda_stream)
sol.comm_init(sol.comm_unique_id(), 0, 1)
print(f"rank 3 of 8: {len(lay.cells_local)} tets, {len(lay.shared_local)} shared nodes, buffer {iface.numel()} doubles")
for route, env in (("eager", "0"), ("graph", "1"), ("eager", "0"), ("graph", "1")):
    sol.set_option("synced_graph", int(env))
    sol.step_synced(300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sol.step_synced(3000)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / 3000
    print(f"  {route:5s}: {us:7.2f} us/step", flush=True)
sol.set_resident_kernel(False)
t = sol.time_steps(1000) * 1e3 / 1000
print(f"  exchange-free, one fused launch per step: {t:7.2f} us/step")
