#!/usr/bin/env python3
"""Diagnostic: every rank's partition of the N = 2, 4, 8 bench configurations on ONE GPU - plan, whether the
resident kernel holds it, and its exchange-free step time (the slowest rank paces a synchronised run).

    python tools/rank_survey.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import N_FOR_GPUS, E, NU, RHO, FZ, ALPHA, GAMMA
import synchronization_avoiding_algorithms_amd as saa
from synchronization_avoiding_algorithms_amd import fem_setup as fs
from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, slab_partition, structured_beam
for world in (2, 4, 8):
    n = N_FOR_GPUS[world]
    mesh = structured_beam(n)
    epart = slab_partition(mesh, world)
    layouts, gshared = fs.build_layouts(mesh.tets, epart, world, len(mesh.points), clamp_nodes(mesh))
    lumped, fpre = fs.lumped_mass_and_load(mesh.points, mesh.tets, RHO, FZ)
    dt = fs.cfl_dt(mesh.points, mesh.tets, E, NU, RHO, GAMMA)
    lmd, mu = fs.lame(E, NU)
    for r in range(world):
        lay = layouts[r]
        sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, lumped[lay.local_dof], fpre[lay.local_dof],
                                    lay.dirichlet_dofs, lmd, mu, dt, ALPHA, shared_local=lay.shared_local,
                                    shared_slots=lay.shared_slots, n_global_shared=len(gshared))
        st, ri = sol.plan_stats(), sol.resident_kernel_info()
        sol.step(200)
        us = sol.time_steps(2000) / 2000 * 1e3
        print(f"N={world} rank {r}: tets {len(lay.cells_local)} nodes {len(lay.nodes)} shared {len(lay.shared_local)} "
              f"blocks {st['n_blocks']} max_owned {st['max_owned']} max_local {st['max_local']} resident {ri['capable']} "
              f"lds {ri['lds_bytes']}  exchange-free step {us:.2f} us", flush=True)
        sol.close()
