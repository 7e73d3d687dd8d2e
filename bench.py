#!/usr/bin/env python3
"""Headline benchmark: explicit linear-tet elastodynamics steps on synthetic cantilevers.

    python bench.py --gpus N --steps K --warmup W

One *step* = one explicit time step of the whole mesh (per partition: the resident multi-step kernel, or one fused
kernel per step; for N > 1 the forces of the shared nodes are summed across ranks every step - by direct xGMI peer
stores inside the step kernel when that path proves itself in a child-process preflight, else by an RCCL all-reduce;
with the peer exchange in use the RCCL variant is measured too and reported as `rccl_allreduce`).  N = 1 runs
BASELINE.json configs[2] (the ~1M-tet beam, one partition); N > 1 keeps ~1M tets per GPU (weak scaling; N = 8 is
configs[3], the ~8M-tet beam in 8 slabs; `sync_avoiding` = configs[4]).  Rank 0 prints ONE JSON line.

Extra objects on the N = 1 line:
  roofline      algorithmic bytes/step (16*Ne + 216*Nn) / HIP-event time of the fused
                kernel's stream, against the 8 TB/s HBM peak of MI355X_MICROARCH.md.
  cpu_baseline  the CPU oracle ("port": SciPy CSR K.dot + the NumPy update, the reference's own per-step
                operations) timed on one host core on a bounded sample; the same sample is stepped on the GPU
                and compared (parity.rel_l2).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

E, NU, RHO, FZ, ALPHA, GAMMA = 1e6, 0.3, 1.0, 0.5, 0.5, 0.9
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
# mesh refinement per GPU count: 150*n^3 tets ~ N * 1.03M
N_FOR_GPUS = {1: 19, 2: 24, 3: 27, 4: 30, 5: 32, 6: 34, 7: 36, 8: 38}


def build_rank_solver(mesh, n_parts, rank, device, block_nodes=0, threads=0):
    """One rank's solver outside a process group (tools, the parity leg): layout of that rank only, set-up fields from
    the HIP kernels; dt from this rank's elements and their neighbours (on the uniform synthetic beams that is the global
    CFL step; inside a process group PartitionedSolver takes the minimum over the ranks)."""
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, slab_partition

    lmd, mu = fs.lame(E, NU)
    epart = slab_partition(mesh, n_parts) if n_parts > 1 else np.zeros(len(mesh.tets), dtype=np.int64)
    lay, gshared, l_M, F_rankwise, dt = fs.rank_problem(mesh.points, mesh.tets, clamp_nodes(mesh), epart, rank, n_parts,
                                                         E, NU, RHO, FZ, GAMMA, device)
    sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, l_M, F_rankwise, lay.dirichlet_dofs, lmd, mu,
                                dt, ALPHA, shared_local=lay.shared_local, shared_slots=lay.shared_slots,
                                n_global_shared=len(gshared), device=device, block_nodes=block_nodes, threads=threads)
    return sol, lay, gshared, dt


def measured_copy_bandwidth(n_bytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + write bytes per second) of this GPU: the practical HBM ceiling
    SURVEY.md section 8(d) asks to report next to the nominal 8 TB/s."""
    import torch

    a = torch.empty(n_bytes // 8, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n_bytes * reps / (e0.elapsed_time(e1) * 1e-3)


def cpu_baseline_and_parity(sample_n=10, steps=6000, parity_steps=3000):
    """Oracle (CPU port of the reference's per-step operations: SciPy CSR ``K.dot`` + the NumPy update expression) timed
    on a bounded sample, on ONE host core (SciPy's SpMV is single-threaded) and on P = min(8, cores) cores the way the
    reference runs distributed (one process per x-slab, shared-node forces summed every step:
    ``oracle/cpu_baseline_mp.py``); the GPU steps the same sample for the parity figure."""
    from oracle import cpu_baseline_mp
    from oracle import fem_oracle as fo
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(sample_n)
    ranks, dt, _, _ = fo.setup_problem(mesh.points, mesh.tets, mesh.triangles, 1,
                                       np.zeros(len(mesh.tets), dtype=int))
    rp = ranks[0]
    d0 = np.zeros((len(rp.local_dof), 1))
    dn = np.zeros_like(d0)
    tn = 0
    t0 = time.perf_counter()
    snap = None
    for i in range(steps):
        d1 = fo.explicit_step(rp.K, rp.F, rp.dirichlet, tn, dt, d0, dn, rp.l_M, ALPHA)
        dn, d0 = d0, d1
        tn = tn + dt
        if i + 1 == parity_steps:
            snap = d0
    cpu_s = time.perf_counter() - t0
    sol, lay, _, gdt = build_rank_solver(mesh, 1, 0, 0)
    assert gdt == dt and np.array_equal(lay.nodes, rp.nodes)
    sol.step(parity_steps)
    g0, _, _ = sol.get_state()
    sol.close()
    rel = float(np.linalg.norm(g0 - snap) / np.linalg.norm(snap))
    ne = len(mesh.tets)
    what = f"synthetic beam n={sample_n} ({ne} tets, {len(mesh.points)} nodes)"
    one = {"value": ne * steps / cpu_s, "unit": "element-updates/s", "cores": 1,
           "sample": f"{what}, {steps} steps, scipy CSR K.dot + numpy update (oracle/fem_oracle.py), {cpu_s:.2f} s"}
    cores = min(8, os.cpu_count() or 1)
    base = dict(one, kind="port")
    if cores > 1:
        try:
            mp_steps = 5 * steps  # ~5-10 s of wall time on 8 cores
            mp = cpu_baseline_mp.run(cores, sample_n, mp_steps)
            base = {"value": ne * mp_steps / mp["seconds"], "unit": "element-updates/s", "cores": cores, "kind": "port",
                    "sample": f"{what} in {cores} x-slabs, one process per slab ({os.cpu_count()} host cores), {mp_steps} "
                              f"steps, per step scipy CSR K.dot + sum of the {mp['n_shared']} shared nodes' forces "
                              f"over the ranks in rank order (shared memory) + numpy update "
                              f"(oracle/cpu_baseline_mp.py), {mp['seconds']:.2f} s",
                    "one_core": one}
        except Exception as e:  # noqa: BLE001 - the one-core figure is still a valid baseline
            base["multi_process_failed"] = str(e)[:200]
    return (base, {"rel_l2": rel, "mesh": f"synthetic beam n={sample_n}", "steps": parity_steps, "tolerance": 1e-10})


def preflight_main(world, rank, local_rank):
    """Child process of one rank (bench.py --preflight): a tiny partitioned problem stepped through the peer exchange
    on the real devices.  Exit code 0 = the direct xGMI path works here; anything else (including a crash of this
    process) makes the parent fall back to the RCCL all-reduce."""
    import torch
    import torch.distributed as dist

    import datetime

    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=90))  # a crashed peer must not park the others
    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver
    from synchronization_avoiding_algorithms_amd.mesh import slab_partition, structured_beam

    mesh = structured_beam(4)
    part = PartitionedSolver(mesh.points, mesh.tets, mesh.triangles, slab_partition(mesh, world), rank, world, E=E, nu=NU,
                             rho=RHO, fz=FZ, alpha=ALPHA, gamma=GAMMA, device=local_rank, exchange="peer")
    ok = part.exchange == "peer"
    if ok:
        part.step_synced(5)      # one launch per step
        part.step_synced(120)    # resident kernel
        d0 = part.get_state()[0]
        ok = bool(np.isfinite(d0).all() and np.abs(d0).max() > 0)
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    part.close()
    dist.destroy_process_group()
    return 0 if all(flags) else 3


def run_preflight(args, world):
    """Runs preflight_main in a child process BEFORE this process touches the GPU; True iff it exited cleanly."""
    import subprocess

    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # the children's rank 0 hosts their own rendezvous store
    env["SAA_PEER_TIMEOUT_S"] = env.get("SAA_PEER_TIMEOUT_S", "20")
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--preflight", "--gpus", str(world)]
        if args.same_device:
            cmd.append("--same-device")
        r = subprocess.run(cmd, env=env, timeout=420, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return r.returncode == 0
    except Exception:  # noqa: BLE001 - timeout, spawn failure
        return False


def sync_avoiding_leg(part, args, rank, world, ne_total, fence):
    """BASELINE.json configs[4] the way the reference's workflow produces it (README.md:33-38), all on the GPUs:
      1. ground truth  - the synchronised run from rest (Data_prepare.py:223-240), recording every rank's shared-dof
                         history (what Shared_extraction.py slices out) and the full state at the end of every window;
      2. training      - one LSTM per rank on ITS history (Model_training.py; training.train_on_history: windowing,
                         [-1, 0] scaling, Adam with 0.998^epoch decay, HIP-graph optimiser step), for a bounded time;
      3. sync-avoiding - the same simulation again (Online_predictor.py:251-318): n_past*filter_size synchronised
                         steps, then windows of n_future*filter_size steps in which the shared dofs come from the
                         rank's model and NO exchange is issued.  Timed: prediction + stepping of the windows.
    Reported next to the throughput: the rel-L2 distance between the sync-avoiding and the synchronised displacement
    field (all nodes, all ranks) at the end of each window."""
    import torch
    import torch.distributed as dist

    from synchronization_avoiding_algorithms_amd import predictor as pr
    from synchronization_avoiding_algorithms_amd import training as tr

    n_p, n_f, n_s, hid, windows = 20, 20, args.sa_filter, 50, args.sa_windows
    warm, win = n_p * n_s, n_f * n_s
    n_truth = max(args.sa_truth_steps, warm + windows * win)
    # the predicted windows are the LAST ones of the recorded run: the synchronised phase of the hybrid loop
    # (Online_predictor.py:253-275) lasts until `start` >= n_past*filter_size.  Right after the minimal warm-up the ramped
    # load has barely moved the beam, and a relative error against a field of ~1e-9 says nothing about the predictor.
    start = n_truth - windows * win
    dev, sol = part.tensor_device, part.solver
    zero = np.zeros(sol.n_dof)
    width = part.input_size
    # 1. ground truth
    sol.set_state(zero, zero, 0.0)
    truth = torch.zeros((n_truth, width), dtype=torch.float64, device=dev)
    marks = [start + (w + 1) * win for w in range(windows)]
    snaps, pos = [], 0
    for m in marks:
        part.step_synced(m - pos, truth, pos)
        pos = m
        snap = torch.empty(sol.n_dof, dtype=torch.float64, device=dev)
        sol.get_state_device(snap, None)
        snaps.append(snap)
    fence()
    # 2. one model per rank, trained on its own history
    t0 = time.perf_counter()
    with torch.enable_grad():
        model, smax, smin, tl, vl = tr.train_on_history(truth, n_s, n_p, n_f, cut_off=1.0, seed=1234 + rank,
                                                        hidden_size=hid, max_seconds=args.sa_train_seconds)
    fence()
    train_s = time.perf_counter() - t0
    groups = truth[::n_s].shape[0] - n_p - n_f + 1
    del truth
    predictor = pr.DevicePredictor(model, n_p, n_f, n_s, smax, smin)
    # 3. the same simulation in sync-avoiding mode
    sol.set_state(zero, zero, 0.0)
    hist = torch.zeros((n_truth, width), dtype=torch.float64, device=dev)
    errs, elapsed, i = [], 0.0, start
    with torch.no_grad():
        part.step_synced(start, hist, 0)
        for _ in range(3):  # untimed: the predictor captures its HIP graph on the third call
            predictor(start, hist)
        for w in range(windows):
            fence()
            t0 = time.perf_counter()
            table = predictor(i, hist)
            part.step_predicted(win, table, 0, hist, i)
            i += win
            fence()
            elapsed += time.perf_counter() - t0
            cur = torch.empty(sol.n_dof, dtype=torch.float64, device=dev)
            sol.get_state_device(cur, None)
            sums = torch.stack([(cur - snaps[w]).square().sum(), snaps[w].square().sum()])
            dist.all_reduce(sums)  # nodes on an interface count once per holder
            errs.append(float(torch.sqrt(sums[0] / sums[1]).item()))
    t = torch.tensor([elapsed, train_s], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, train_s = (float(v) for v in t.tolist())
    finite = torch.tensor([float(torch.isfinite(hist[-1]).all())], device="cuda")
    dist.all_reduce(finite, op=dist.ReduceOp.MIN)
    return {"value": ne_total * windows * win / elapsed, "unit": "element-updates/s",
            "ms_per_step": 1e3 * elapsed / (windows * win), "steps": windows * win, "windows": windows,
            "n_past": n_p, "n_future": n_f, "filter_size": n_s, "input_size_rank0": width,
            "synchronised_steps_before": start,
            "rel_l2_vs_synchronised": errs, "state_finite": bool(finite.item()),
            "training": {"truth_steps": n_truth, "windows": int(groups), "epochs": len(tl), "seconds": train_s,
                         "train_mse_first_last": [tl[0], tl[-1]], "validation_mse_last": vl[-1], "hidden_size": hid},
            "note": f"after {start} synchronised steps every rank's LSTM (trained in this run on the synchronised "
                    f"history of its own shared dofs) predicts them for {win}-step windows; no collective inside a "
                    "window; predictor time included; rel_l2_vs_synchronised = whole displacement field against the "
                    "synchronised run at the end of each window"}


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks (one process per GPU) through
    ``torch.distributed.run`` - the reference's whole launch story is ``mpirun -np P python3 ...``
    (/root/reference README.md:33-38) - BEFORE this process makes any GPU call (it never does), relay rank 0's single
    JSON line and exit non-zero if the ranks failed or printed no line."""
    import signal
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the peer exchange need it on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    line = None
    try:
        for ln in proc.stdout:  # rank 0's JSON line is the only thing the ranks write to stdout
            try:
                if ln.lstrip().startswith("{") and "metric" in json.loads(ln):
                    line = ln.strip()
                    continue
            except ValueError:
                pass
            sys.stderr.write(ln)
        rc = proc.wait(timeout=args.launch_timeout)
    except BaseException:  # timeout, Ctrl-C: end exactly the process group started above
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        raise
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        raise SystemExit(rc if rc != 0 else 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--refine", type=int, default=0, help="override mesh refinement n (25n x n x n cubes)")
    ap.add_argument("--block-nodes", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--no-sync-avoiding", action="store_true", help="N > 1: skip the sync-avoiding-mode leg")
    ap.add_argument("--sa-windows", type=int, default=3, help="sync-avoiding leg: predicted windows that are timed")
    ap.add_argument("--sa-filter", type=int, default=150, help="sync-avoiding leg: filter_size n_s (Online_predictor.py:59)")
    ap.add_argument("--sa-truth-steps", type=int, default=30000,
                    help="sync-avoiding leg: synchronised steps recorded as training data")
    ap.add_argument("--sa-train-seconds", type=float, default=60.0, help="sync-avoiding leg: training time bound per rank")
    ap.add_argument("--no-rccl-leg", action="store_true",
                    help="N > 1: skip the extra measurement with the RCCL all-reduce when the peer exchange is in use")
    ap.add_argument("--torch-exchange", action="store_true",
                    help="N > 1: all-reduce through torch.distributed (same as --exchange torch)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer", "rccl", "torch"],
                    help="N > 1: how shared-node forces travel: peer = direct xGMI stores (saa_step_peer), rccl = "
                         "ncclAllReduce from C++, torch = torch.distributed.all_reduce; auto tries them in that order")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--preflight", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="the timed call of --steps steps is repeated until the timed region lasts at least this long")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.same_device:
        local_rank = 0
    if args.preflight:
        raise SystemExit(preflight_main(world, rank, local_rank))
    # N > 1: the direct peer exchange maps other processes' device memory - first use on this machine happens in a
    # child process, so that a fault there costs the child, not the benchmark (which then takes the RCCL all-reduce)
    exchange = "torch" if args.torch_exchange else args.exchange
    peer_ok = True
    if world > 1 and exchange == "auto" and (not args.same_device or os.environ.get("SAA_BENCH_FORCE_PREFLIGHT")):
        peer_ok = run_preflight(args, world)

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver
    from synchronization_avoiding_algorithms_amd.mesh import slab_partition, structured_beam

    if world > 1 and exchange == "auto":  # every rank takes the same transport
        flags = [None] * world
        dist.all_gather_object(flags, bool(peer_ok))
        if not all(flags):
            exchange = "rccl"
    preflight = None if world == 1 or args.exchange != "auto" or args.torch_exchange else bool(peer_ok)
    n = args.refine or N_FOR_GPUS.get(world, int(round((world * 1028850 / 150.0) ** (1 / 3))))
    mesh = structured_beam(n)
    ne_total, nn_total = len(mesh.tets), len(mesh.points)
    epart = slab_partition(mesh, world) if world > 1 else np.zeros(ne_total, dtype=np.int64)
    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build_and_time(how):
        """Partition solver with the given transport, W warm-up and K timed steps; ``ok`` is False on any rank if a
        wait inside a kernel gave up (peer exchange: a neighbour's values never arrived)."""
        part = PartitionedSolver(mesh.points, mesh.tets, mesh.triangles, epart, rank, world, E=E, nu=NU, rho=RHO,
                                 fz=FZ, alpha=ALPHA, gamma=GAMMA, device=local_rank, block_nodes=args.block_nodes,
                                 threads=args.threads, exchange=how)
        part.step_synced(args.warmup)  # world == 1: plain steps; else one exchange of shared-node forces per step
        # The timed call is `part.step_synced(args.steps)`.  Which kernels that runs depends on the call length (calls
        # of >= 8 steps take the resident kernel: one cooperative launch), so exactly that call is issued once more
        # untimed - the first launch of a kernel pays code-object upload and cooperative-launch set-up - and its
        # duration sizes the number of timed repetitions so that the timed region lasts >= --min-timed-ms.
        fence()
        t0 = time.perf_counter()
        part.step_synced(args.steps)
        fence()
        est = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([est], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            est = float(t.item())
        calls = int(min(max(1, np.ceil(args.min_timed_ms * 1e-3 / max(est, 1e-7))), 100000))
        for attempt in range(4):  # (the untimed call is slower than the warm ones: re-size until the region is long enough)
            fence()
            t0 = time.perf_counter()
            for _ in range(calls):
                part.step_synced(args.steps)
            fence()
            total = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([total], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                total = float(t.item())
            if total >= args.min_timed_ms * 1e-3 or attempt == 3:
                break
            calls = int(min(np.ceil(1.15 * calls * args.min_timed_ms * 1e-3 / max(total, 1e-7)), 100000))
        elapsed = total / calls  # per call of args.steps steps
        build_and_time.calls = calls
        ok = True
        try:
            part.solver.synchronize()  # raises if a bounded wait timed out
        except RuntimeError:
            ok = False
        if world > 1:
            flags = [None] * world
            dist.all_gather_object(flags, ok)
            ok = all(flags)
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return part, elapsed, ok

    part, elapsed, ok = build_and_time(exchange)
    timed_calls = build_and_time.calls
    retried = None
    if not ok and world > 1 and part.exchange == "peer":  # never seen; costs one rebuild if it ever happens
        retried = "peer exchange timed out during the timed run; measured again with the RCCL all-reduce"
        part.close()
        part, elapsed, ok = build_and_time("rccl")
        timed_calls = build_and_time.calls
    if not ok:
        raise SystemExit("bench: a wait inside the step kernels timed out")
    sol, gshared, dt = part.solver, part.global_shared, part.dt

    # N > 1: the same partitions in sync-avoiding mode (BASELINE.json configs[4]; Online_predictor.py:251-318).
    sync_avoiding = None
    if world > 1 and not args.no_sync_avoiding:
        sync_avoiding = sync_avoiding_leg(part, args, rank, world, ne_total, fence)

    out = None
    if rank == 0:
        stats = sol.plan_stats()
        out = {
            "metric": "element_updates_per_s", "value": ne_total * args.steps / elapsed,
            "unit": "element-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_calls": timed_calls,
            "ms_per_step": 1e3 * elapsed / args.steps, "steps_per_s": args.steps / elapsed,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"synthetic 25n x n x n Kuhn-tet cantilever n={n}: {ne_total} tets, "
                                   f"{nn_total} nodes, {world} x-slab partition(s), fp64, E=1e6 nu=0.3 "
                                   f"alpha=0.5 ramped body force, dt={dt:.6e}",
                       "exchange": "none (1 partition)" if world == 1 else
                                   {"peer": f"every step the fp64 forces of the shared nodes ({len(gshared)} in all) are "
                                            "stored into the neighbour ranks' memory (HIP IPC, xGMI peer stores) and "
                                            "summed in rank order (saa_step_peer); no collective",
                                    "rccl": f"all-reduce of {3 * len(gshared)} fp64 shared-node forces every step, "
                                            "ncclAllReduce issued from C++ (saa_step_synced)",
                                    "torch": f"all-reduce of {3 * len(gshared)} fp64 shared-node forces every step, "
                                             f"torch.distributed ({args.backend})"}[part.exchange],
                       "plan": stats},
        }
    if rank == 0 and sync_avoiding is not None:
        out["sync_avoiding"] = sync_avoiding
    if rank == 0 and world > 1:
        out["config"]["peer_preflight_rank0"] = preflight  # child-process trial of the peer exchange (None: not run)
        if retried:
            out["config"]["retried"] = retried
    if world == 1:
        # roofline of the dominant (only) kernel: HIP events on the kernel's own stream
        # one launch of the resident kernel advances `spl` steps (one launch = spl * Ne element-updates); without it
        # (plan does not fit LDS / not all workgroups co-resident) one launch of the fused kernel is one step
        res = sol.resident_kernel_info()
        spl = res["steps_per_launch"] if res["capable"] else 1
        launches = 12 if spl > 1 else 3000
        sol.time_steps(2 * spl if spl > 1 else 200)  # settle the clocks on this very path
        ms = sol.time_steps(launches * spl)
        b_alg = 16 * ne_total + 216 * nn_total
        achieved = b_alg * launches * spl / (ms * 1e-3)
        out["roofline"] = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK, "traffic": None,
                           "kernel": "persistent_steps_kernel<false,false>" if spl > 1 else "fused_step_kernel<false>",
                           "steps_per_launch": spl, "launches_timed": launches, "avg_launch_us": 1e3 * ms / launches,
                           "us_per_step": 1e3 * ms / (launches * spl),
                           "algorithmic_bytes_per_step": b_alg, "algorithmic_bytes_per_launch": b_alg * spl,
                           "algorithmic_bytes_per_element_update": b_alg / ne_total,
                           "note": "algorithmic bytes = SURVEY.md section 8(d) figure for a kernel that re-reads the "
                                   "partition every step; the resident kernel keeps it in LDS and moves less "
                                   "(see traffic)"}
        # HBM traffic per launch from the committed PMC passes of this same kernel and mesh (tools/pmc_collect.sh:
        # rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, nothing else enabled; KiB units; FETCH_SIZE
        # doubled: gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md "HBM").  Only for the meshes that were
        # profiled (n = 19 resident, n = 38 fused); otherwise null.
        try:
            with open(os.path.join(REPO, "profiles", "r02_pmc_summary.json")) as fh:
                pmc = json.load(fh)
            key = f"r02_resident{n}" if spl > 1 else f"r02_fused{n}"
            fetch = pmc[f"{key}:FETCH_SIZE"]["mean_per_dispatch"]
            write = pmc[f"{key}:WRITE_SIZE"]["mean_per_dispatch"]
            out["roofline"]["traffic"] = (2.0 * fetch + write) * 1024.0
            out["roofline"]["traffic_source"] = ("profiles/r02_pmc_summary.json (rocprofv3 --pmc, separate passes, "
                                                 "launches of the same length)")
        except (OSError, KeyError, ValueError):
            pass
        copy_bw = measured_copy_bandwidth()
        out["roofline"]["measured_copy_GBps"] = copy_bw / 1e9
        out["roofline"]["frac_of_measured_copy"] = achieved / copy_bw
        if not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity()
    sol.close()
    # N > 1: BASELINE.json configs[3] names the RCCL all-reduce as the per-step exchange.  When `value` above was
    # measured with the peer exchange, the same partitions are stepped once more with ncclAllReduce issued from C++
    # (saa_step_synced) and reported next to it.
    leg_ok = (args.backend == "nccl" and not args.same_device) or bool(os.environ.get("SAA_BENCH_FORCE_RCCL_LEG"))
    if world > 1 and part.exchange == "peer" and leg_ok and not args.no_rccl_leg:
        import threading

        leg_done = threading.Event()

        def watchdog():  # a hung collective must not cost the headline line: after 4 minutes print what has been
            if not leg_done.wait(240):  # measured and leave with a NON-ZERO status, so that the hang is seen
                if rank == 0:
                    out["rccl_allreduce"] = {"value": None, "exchange": "timed out after 240 s"}
                    print(json.dumps(out), flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        k_r, w_r = min(args.steps, 2000), min(args.warmup, 200)
        args_steps, args_warmup = args.steps, args.warmup
        args.steps, args.warmup = k_r, w_r
        try:
            part_r, elapsed_r, ok_r = build_and_time("rccl" if args.backend == "nccl" else "torch")
            how = part_r.exchange
            part_r.close()
        except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline line
            ok_r, how, elapsed_r = False, f"failed: {e}"[:200], float("nan")
        args.steps, args.warmup = args_steps, args_warmup
        if rank == 0:
            out["rccl_allreduce"] = ({"value": ne_total * k_r / elapsed_r, "unit": "element-updates/s",
                                      "ms_per_step": 1e3 * elapsed_r / k_r, "steps": k_r, "warmup": w_r, "exchange": how,
                                      "note": "same partitions; shared-node forces summed by an all-reduce of "
                                              f"{3 * len(gshared)} doubles every step instead of the peer exchange"}
                                     if ok_r else {"value": None, "exchange": how})
        leg_done.set()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("SAA_BENCH_DUMP_MAPS"):  # profiling runs: lets a crash in an exit handler be attributed
        with open("/proc/self/maps") as src, open(os.environ["SAA_BENCH_DUMP_MAPS"], "w") as dst:
            dst.write(src.read())


if __name__ == "__main__":
    main()
