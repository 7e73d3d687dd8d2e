"""Misbehaviour injected into bench.py by the tests of its deadline handling (``SAA_BENCH_HOOKS=tests.bench_hooks``):
which point of the run misbehaves, and how, comes from ``SAA_BENCH_HOOK_PLAN`` - "preflight=hang", "preflight=fail",
"rccl_leg=hang", comma-separated.  Nothing of this lives in bench.py itself."""
import os
import time


def hook(point):
    plan = dict(item.split("=", 1) for item in os.environ.get("SAA_BENCH_HOOK_PLAN", "").split(",") if "=" in item)
    what = plan.get(point)
    if what == "hang":
        time.sleep(3600)
    elif what == "fail":
        raise SystemExit(3)
