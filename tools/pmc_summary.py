#!/usr/bin/env python3
"""Folds rocprofv3 --pmc outputs into profiles/r02_pmc_summary.json (--out=<file> for another).

    python tools/pmc_summary.py <key> <rocprof output dir> [<kernel name substring>] [--grid=G] [--out=profiles/x.json]

For every counter in the directory's *_counter_collection.csv files: mean value per dispatch of the kernels whose
name contains the substring (default: "step").  With --grid G only dispatches of that grid size (threads) count,
which separates the bench mesh from the small meshes other legs of the same command run.
Entry name: "<key>:<COUNTER>".  With --plan-from=<log of the profiled bench run> the plan statistics and the digest of
the kernel sources that run printed (config.plan, config.kernel_sources) are stored as "<key>:plan": bench.py hands the
counters out only for a run with that very plan.
"""
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    key, root = args[0], args[1]
    sub = args[2] if len(args) > 2 else "step"
    grid = None
    full_only = "--full-only" in sys.argv[1:]  # keep only dispatches within 10 % of the largest value of their counter
    for a in sys.argv[1:]:                     # (drops the resident kernel's census launch and short warm-up launches)
        if a.startswith("--grid="):
            grid = int(a.split("=", 1)[1])
    acc = {}
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if sub not in row["Kernel_Name"]:
                    continue
                if grid is not None and int(row["Grid_Size"]) != grid:
                    continue
                rows.append((row["Counter_Name"], float(row["Counter_Value"]), row["Kernel_Name"][:80]))
    top = {}
    for name, value, _ in rows:
        top[name] = max(top.get(name, 0.0), value)
    for name, value, kernel in rows:
        if full_only and value < 0.9 * top[name]:
            continue
        d = acc.setdefault(name, {"sum": 0.0, "n": 0, "kernel": kernel})
        d["sum"] += value
        d["n"] += 1
    out_path = os.path.join(REPO, "profiles", "r02_pmc_summary.json")
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            out_path = os.path.join(REPO, a.split("=", 1)[1])
    try:
        with open(out_path) as fh:
            summary = json.load(fh)
    except OSError:
        summary = {}
    for name, d in acc.items():
        summary[f"{key}:{name}"] = {"dispatches": d["n"], "mean_per_dispatch": d["sum"] / max(d["n"], 1),
                                    "kernel": d["kernel"]}
        print(f"{key}:{name}", summary[f"{key}:{name}"])
    for a in sys.argv[1:]:
        if a.startswith("--plan-from="):
            with open(a.split("=", 1)[1]) as fh:
                lines = [ln for ln in fh if ln.lstrip().startswith("{") and '"metric"' in ln]
            cfg = json.loads(lines[-1])["config"]
            summary[f"{key}:plan"] = dict(cfg["plan"], kernel_sources=cfg.get("kernel_sources"))
            print(f"{key}:plan", summary[f"{key}:plan"])
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
