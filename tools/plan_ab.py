#!/usr/bin/env python3
"""Diagnostic: the plan's second-list layout (SAA_PLAN_FIRST_ROUND_CAP), chunk repair (SAA_PLAN_CHUNK_REPAIR) and cuts along
layers of the mesh size (SAA_PLAN_SNAP_CUTS) on / off for one rank of a bench configuration, same box, alternating:   python tools/plan_ab.py <world> <rank> [rounds]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _diag import diag_library_path  # noqa: E402

DIAG = diag_library_path()
world, rank = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
CHILD = r"""
import sys, json
sys.path.insert(0, %r)
sys.argv = ['bench.py']
from bench import N_FOR_GPUS, build_rank_solver
from synchronization_avoiding_algorithms_amd.mesh import structured_beam
sol = build_rank_solver(structured_beam(N_FOR_GPUS[%d]), %d, %d, 0)[0]
sol.time_steps(2000)
us = [1e3 * sol.time_steps(4000) / 4000 for _ in range(3)]
print(json.dumps({'us': us, 'plan': sol.plan_stats()}))
""" % (REPO, world, world, rank)
for r in range(rounds):
    for cap, rep, snap in (("0", "0", "0"), ("-", "-", "-")):  # everything off / the defaults
        env = dict(os.environ, SAA_LIB_PATH=DIAG)  # the SAA_PLAN_* switches exist in the diagnostic build only
        if cap != "-":
            env.update(SAA_PLAN_FIRST_ROUND_CAP=cap, SAA_PLAN_CHUNK_REPAIR=rep, SAA_PLAN_SNAP_CUTS=snap)
        p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
        if p.returncode:
            print(p.stderr[-800:])
            continue
        d = json.loads(p.stdout.strip().splitlines()[-1])
        print(f"N={world} rank {rank} cap {cap} repair {rep} snap {snap}: " + " ".join(f"{u:.3f}" for u in d["us"]) +
              f" us/step; items {d['plan']['n_items']} halo {d['plan']['n_halo_total']} conflict {d['plan']['lds_conflict_factor']:.3f} / {d['plan']['lds_atomic_conflict_factor']:.3f}", flush=True)
