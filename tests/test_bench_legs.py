"""bench.py's deadline book-keeping (no GPU): once the headline exists the line is printed whatever a later leg does."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, time
sys.path.insert(0, {repo!r})
sys.argv = ["bench.py"]
import bench
legs = bench.Legs(time.time(), {budget}, 0)
legs.begin("headline"); legs.end("headline")
with legs.lock:
    legs.line = {{"metric": "element_updates_per_s", "value": 1.0}}
legs.skip("sync_avoiding", "test")
legs.begin("rccl_allreduce", limit_s={limit})
{body}
legs.end("rccl_allreduce")
legs.emit(final=True)
"""


def _run(budget, limit, body):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", SCRIPT.format(repo=REPO, budget=budget, limit=limit, body=body)],
                       capture_output=True, text=True, timeout=120)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, lines, time.time() - t0


def test_a_leg_that_finishes_gives_one_line_and_status_zero():
    rc, lines, _ = _run(60, 30, "time.sleep(0.1)")
    assert rc == 0 and len(lines) == 1
    assert lines[0]["legs"] == {"headline": "done", "sync_avoiding": "skipped: test", "rccl_allreduce": "done"}
    assert lines[0]["value"] == 1.0 and "total" in lines[0]["leg_seconds"] and lines[0]["budget_s"] == 60


def test_a_leg_over_its_limit_is_cut_off_with_the_line_printed():
    rc, lines, wall = _run(60, 1.5, "time.sleep(600)")
    assert rc == 3 and len(lines) == 1 and wall < 20
    assert lines[0]["legs"]["rccl_allreduce"] == "unfinished: leg time limit" and lines[0]["legs"]["headline"] == "done"


def test_the_global_deadline_prints_the_line_before_it_passes():
    rc, lines, wall = _run(11, "None", "time.sleep(600)")  # (the line leaves 8 s before the deadline)
    assert rc == 3 and len(lines) == 1 and wall < 11
    assert lines[0]["legs"]["rccl_allreduce"] == "unfinished: deadline"


def test_job_start_time_ignores_shells_and_test_runners():
    """The budget counts from the start of the job's own processes (bench.py, torch.distributed.run), not from a shell
    or test runner that happens to carry 'bench.py' on its command line."""
    code = ("import sys, time; sys.path.insert(0, %r); sys.argv=['bench.py']; import bench; "
            "print(time.time() - bench.job_start_time())" % REPO)
    # the parent of that interpreter is this pytest process (started long ago, 'bench.py' appears in no argv[1])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60,
                       env={k: v for k, v in os.environ.items() if k != "SAA_BENCH_T0"})
    assert r.returncode == 0 and 0 <= float(r.stdout) < 30, (r.stdout, r.stderr)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60,
                       env=dict(os.environ, SAA_BENCH_T0=repr(time.time() - 100)))
    assert 99 < float(r.stdout) < 130


def test_committed_counters_are_only_handed_out_for_the_profiled_plan():
    """roofline.traffic / onchip come from committed PMC passes of ANOTHER run of the same command: bench.py prints them only
    while the plan statistics of this run equal the ones recorded with the counters, and says why otherwise."""
    import importlib

    sys.argv = ["bench.py"]
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    with open(os.path.join(REPO, "profiles", "r04_pmc_summary.json")) as fh:
        rec = json.load(fh)["r04_resident19:plan"]
    stats = {k: rec[k] for k in bench.PLAN_IDENTITY}
    traffic, onchip, src, note = bench.committed_counters("resident19", stats)
    assert traffic > 1e9 and onchip is not None and "r04_pmc_summary.json" in src
    assert 0.5 < onchip["valu_busy"] < 1.0 and 0.5 < onchip["lds_busy"] < 1.0
    other = dict(stats, n_items=stats["n_items"] + 1)  # any other plan: refused, with the difference named
    traffic, onchip, src, note = bench.committed_counters("resident19", other)
    assert traffic is None and onchip is None and "n_items" in note and "profile again" in note
    assert bench.committed_counters("resident7", stats) == (None, None, None, None)  # a mesh that was never profiled
