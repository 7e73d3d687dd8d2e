#!/usr/bin/env python3
"""Diagnostic: partitions around the capacity of the resident kernel (one block per CU up to 840 owned nodes on average):
Delaunay beams with 1.10x and 1.13x the node count of the 1M-tet bench mesh - resident or not, plan, us per step."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
sys.argv=['bench.py']
from bench import build_rank_solver
from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam, structured_beam
for name, mesh in (("delaunay 1.10", delaunay_beam(19, density=1.10)), ("delaunay 1.13", delaunay_beam(19, density=1.13))):
    sol = build_rank_solver(mesh, 1, 0, 0)[0]
    st, ri = sol.plan_stats(), sol.resident_kernel_info()
    sol.time_steps(4000)
    us = [1e3 * sol.time_steps(20000) / 20000 for _ in range(2)]
    print(name, len(mesh.points), len(mesh.tets), "resident", ri, "us/step", us, {k: st[k] for k in ("n_blocks","threads","max_owned","max_local","n_items","lds_bytes")}, flush=True)
    sol.close()
