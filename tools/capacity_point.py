#!/usr/bin/env python3
"""Diagnostic: partitions around and beyond the capacity of the resident kernel (one block per CU up to 840 owned nodes on
average): Delaunay beams with 1.10x ... 1.8x the node count of the 1M-tet bench mesh - resident or not, plan, us per step;
beyond the capacity the automatic plan (512-thread blocks of ~720 nodes) against ONE 1024-thread block per CU."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py"] + sys.argv[1:]
from bench import build_rank_solver  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam  # noqa: E402

for density in (1.13, 1.8, 2.4):
    mesh = delaunay_beam(19, density=density)
    for bn, th in ((0, 0), (720, 512)):
        try:
            sol = build_rank_solver(mesh, 1, 0, 0, bn, th)[0]
        except Exception as e:  # noqa: BLE001
            print(f"density {density} block_nodes {bn} threads {th}: {str(e)[:120]}", flush=True)
            continue
        st, ri = sol.plan_stats(), sol.resident_kernel_info()
        sol.time_steps(3000)
        us = [1e3 * sol.time_steps(10000) / 10000 for _ in range(2)]
        print(f"density {density}: {len(mesh.points)} nodes {len(mesh.tets)} tets, block_nodes {bn or 'auto'} threads {th or 'auto'}: resident "
              f"{ri['capable']}, {us[0]:.2f} {us[1]:.2f} us/step = {us[1] * 1e6 / len(mesh.tets):.2f} us per Mtet; "
              + str({k: st[k] for k in ("n_blocks", "threads", "max_owned", "max_local", "n_items", "lds_bytes")}), flush=True)
        sol.close()
