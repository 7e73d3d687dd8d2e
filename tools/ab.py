#!/usr/bin/env python3
"""A/B of step-kernel builds on one GPU box: every library given on the command line steps the same meshes in the same
process order, alternating, so that clocks and box are the same for all of them.

    python tools/ab.py name=path/to/lib.so [name=...] [--cases=resident19,fused19,fused38] [--rounds=3]

Cases: resident<n> (saa_step through the resident kernel, 1000-step launches), resident20x<n> (the same in calls of 20
steps, wall clock), fused<n> (one launch per step), predicted<n> (slab 3 of 8 of the n-beam in windows of 50 predicted
steps, saa_step_predicted, wall clock; predictednohist<n>: without the history record; predictedplain<n>: plain steps
in the same windows).
Each (library, case) runs in a child process (a library is loaded once per process); prints us/step per round and the
median."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, os
sys.path.insert(0, %r)
sys.argv = ['bench.py']
import numpy as np
from bench import build_rank_solver, bench_mesh
case, kind = sys.argv_case
n = int(''.join(ch for ch in case.split('x')[-1] if ch.isdigit()))
mesh = bench_mesh(n, kind)
predicted = case.startswith('predicted')  # BASELINE config 4's per-GPU workload: slab 3 of 8, windows of 50 predicted steps
sol, lay, _, _ = build_rank_solver(mesh, 8 if predicted else 1, 3 if predicted else 0, 0)
resident = case.startswith('resident') or predicted
if not resident:
    sol.set_resident_kernel(False)
assert sol.resident_kernel_info()['capable'] == resident, sol.resident_kernel_info()
import time
if case.startswith('resident20x'):  # calls of 20 steps (the driver's --steps 20): one resident launch per call, wall clock
    def run():
        sol.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            sol.step(20)
        sol.synchronize()
        return 1e6 * (time.perf_counter() - t0) / 8000
    run()
    out = [run() for _ in range(3)]
elif predicted:
    import torch
    w = 3 * len(lay.shared_local)
    table = torch.from_numpy(np.random.default_rng(0).uniform(-1e-6, 1e-6, size=(50, w))).cuda()
    hist = torch.zeros((50, w), dtype=torch.float64, device='cuda')
    def run():
        sol.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            if 'plain' in case:      # predictedplain<n>: the same slab and window length, exchange-free plain steps
                sol.step(50)
            else:                    # predictednohist<n>: the overwrite without the history record
                sol.step_predicted(50, table, 0, None if 'nohist' in case else hist, 0)
        sol.synchronize()
        return 1e6 * (time.perf_counter() - t0) / 10000
    run()
    out = [run() for _ in range(3)]
    assert 'plain' in case or 'nohist' in case or torch.equal(hist, table)
else:
    steps = 4000 if resident else (1000 if n < 30 else 300)
    sol.time_steps(steps)
    out = [1e3 * sol.time_steps(steps) / steps for _ in range(3)]
d0 = sol.get_state()[0]
print(json.dumps({'us': out, 'norm': float(np.linalg.norm(d0)), 'plan': sol.plan_stats()}))
"""


def main():
    libs, cases, rounds, kind = [], ["resident19", "fused19", "fused38"], 2, "structured"
    for a in sys.argv[1:]:
        if a.startswith("--cases="):
            cases = a.split("=", 1)[1].split(",")
        elif a.startswith("--rounds="):
            rounds = int(a.split("=", 1)[1])
        elif a.startswith("--mesh="):
            kind = a.split("=", 1)[1]
        else:
            name, path = a.split("=", 1)
            libs.append((name, os.path.abspath(path)))
    res = {}
    for case in cases:
        for r in range(rounds):
            for name, path in libs:
                env = dict(os.environ, SAA_LIB_PATH=path, SAA_ALLOW_OLD_ABI="1")
                code = CHILD.replace("sys.argv_case", repr((case, kind))) % REPO
                p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
                if p.returncode != 0:
                    print(f"{case} {name}: FAILED\n{p.stderr[-1500:]}", flush=True)
                    continue
                rec = json.loads(p.stdout.strip().splitlines()[-1])
                res.setdefault((case, name), []).extend(rec["us"])
                print(f"{case:12s} {name:10s} round {r}: " + " ".join(f"{u:8.3f}" for u in rec["us"]) +
                      f"  us/step   |d|={rec['norm']:.12e}  conflict {rec['plan']['lds_conflict_factor']:.3f}", flush=True)
    print()
    for (case, name), us in res.items():
        us = sorted(us)
        print(f"{case:12s} {name:10s} median {us[len(us) // 2]:8.3f}  min {us[0]:8.3f} us/step")


if __name__ == "__main__":
    main()
