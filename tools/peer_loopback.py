#!/usr/bin/env python3
"""Diagnostic: cost of the in-kernel peer exchange (push + collect phases of fused_step_kernel<PEER>) on ONE GPU.

Takes the partition a middle rank holds in the driver's N-GPU bench (x-slab of the 25n x n x n beam, two
interfaces), attaches it to imaginary neighbours living in its own inbox (saa_peer_attach_loopback) and times
saa_step_peer against plain saa_step on the same partition.  Local memory latency stands in for xGMI's.

    python tools/peer_loopback.py [--gpus 8] [--rank 3] [--steps 2000]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import use_diag_library  # noqa: E402

use_diag_library()  # diagnostic build of the library; the product .so has none of the saa_debug_* entry points

from bench import ALPHA, E, GAMMA, N_FOR_GPUS, NU, RHO, FZ  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--refine", type=int, default=0)
    ap.add_argument("--block-nodes", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--plain-only", action="store_true")
    args = ap.parse_args()
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import _lib, fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, slab_partition, structured_beam

    from bench import build_rank_solver

    n = args.refine or N_FOR_GPUS[args.gpus]
    mesh = structured_beam(n)

    def make():
        sol, lay, _, _ = build_rank_solver(mesh, args.gpus, args.rank, 0, args.block_nodes, args.threads)
        make.lay = lay
        return sol

    lib = _lib.load()
    lib.saa_debug_time_peer.restype = C.c_int
    plain = make()
    lay = make.lay
    print(f"rank {args.rank} of {args.gpus}: {len(lay.cells_local)} tets, {len(lay.nodes)} nodes, "
          f"{len(lay.shared_local)} shared nodes", flush=True)
    plain.step(20000)  # clocks settle over the first tens of milliseconds of load
    base = min(plain.time_steps(args.steps) for _ in range(3)) / args.steps * 1e3
    print(f"plain step (no exchange)        : {base:7.2f} us/step   plan {plain.plan_stats()}", flush=True)
    plain.close()
    for world in (() if args.plain_only else (2, 3)):
        sol = make()
        sol.peer_attach_loopback(world)
        ms = C.c_double()
        _lib.check(lib.saa_debug_time_peer(sol._h, C.c_int32(20000), C.byref(ms)))
        best = 1e300
        for _ in range(3):
            _lib.check(lib.saa_debug_time_peer(sol._h, C.c_int32(args.steps), C.byref(ms)))
            best = min(best, ms.value)
        us = best / args.steps * 1e3
        print(f"peer step, {world - 1} loopback neighbour(s): {us:7.2f} us/step   (+{us - base:.2f} us)", flush=True)
        sol.close()


if __name__ == "__main__":
    main()
