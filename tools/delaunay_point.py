#!/usr/bin/env python3
"""Diagnostic: the Delaunay beam (mesh.delaunay_beam(19): 190 400 nodes, 1 002 482 tets, no lattice structure) stepped by the
resident kernel with the plan's idle-lane allowance (SAA_PLAN_PAD, diagnostic build) at several values - what a bank clash
costs against an idle lane once the numbering is decided while the groups are formed (saa_plan.cpp: joint_pack_list).

    python tools/delaunay_point.py [pad ...]        (default: 0 0.02 0.04 0.08)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from _diag import diag_library_path  # noqa: E402

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
sys.argv = ['bench.py']
from bench import build_rank_solver
from synchronization_avoiding_algorithms_amd.mesh import Mesh
z = np.load(%r)
mesh = Mesh(z['points'], {'tetra': z['tets'], 'triangle': z['tri']})
sol = build_rank_solver(mesh, 1, 0, 0)[0]
sol.time_steps(4000)
us = [1e3 * sol.time_steps(20000) / 20000 for _ in range(3)]
print(json.dumps({'us': us, 'plan': sol.plan_stats(), 'resident': sol.resident_kernel_info()['capable']}))
"""


def main():
    from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam

    pads = sys.argv[1:] or ["0", "0.02", "0.04", "0.08"]
    mesh = delaunay_beam(19)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "mesh.npz")
        np.savez(path, points=mesh.points, tets=mesh.tets, tri=mesh.triangles)
        diag = diag_library_path()
        for rnd in range(2):  # (clocks differ from minute to minute on a box: every value twice, alternating)
            for pad in pads:
                env = dict(os.environ, SAA_LIB_PATH=diag, SAA_PLAN_PAD=pad)
                p = subprocess.run([sys.executable, "-c", CHILD % (REPO, path)], capture_output=True, text=True, env=env, timeout=600)
                if p.returncode:
                    print(p.stderr[-800:])
                    continue
                d = json.loads(p.stdout.strip().splitlines()[-1])
                st = d["plan"]
                print(f"pad {pad:>5s}: " + " ".join(f"{u:.3f}" for u in d["us"]) + f" us/step  resident {d['resident']}  item slots "
                      f"{st['n_items']}  conflict factors {st['lds_conflict_factor']:.3f} / {st['lds_atomic_conflict_factor']:.3f}", flush=True)


if __name__ == "__main__":
    main()
