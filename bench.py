#!/usr/bin/env python3
"""Headline benchmark: explicit linear-tet elastodynamics steps on synthetic cantilevers.

    python bench.py --gpus N --steps K --warmup W

One *step* = one explicit time step of the whole mesh (per partition: the resident multi-step kernel, or one fused
kernel per step; for N > 1 the forces of the shared nodes are summed across ranks every step - by direct xGMI peer
stores inside the step kernel when that path proves itself in a child-process preflight, else by an RCCL all-reduce;
with the peer exchange in use the RCCL variant is measured too and reported as `rccl_allreduce`).  N = 1 runs
BASELINE.json configs[2] (the ~1M-tet beam, one partition); N > 1 keeps ~1M tets per GPU (weak scaling; N = 8 is
configs[3], the ~8M-tet beam in 8 slabs; `sync_avoiding` = configs[4]).  Rank 0 prints ONE JSON line.

The run has ONE deadline (`--budget-s`, counted from the start of the job) and every leg a budget inside it; the
headline is measured first, and once it exists the line is printed no matter what the later legs do - a leg that does
not fit is skipped, a leg that hangs is cut off by the watchdog, which prints the line with that leg marked and leaves
with a non-zero status (`legs`, `leg_seconds`).

Extra objects on the N = 1 line:
  roofline      contract figure: algorithmic bytes/step (16*Ne + 216*Nn) / HIP-event time on the
                kernel's own stream, against the 8 TB/s HBM peak of MI355X_MICROARCH.md - an EQUIVALENT bandwidth (what a
                kernel re-reading the partition every step would have to sustain), over a region of >= 1 s (`sustained`)
                next to the short region earlier rounds quoted; next to it what the kernel really moves (`traffic`,
                `hbm_measured`) and what really bounds it (`bound`, `onchip`: VALU / LDS / barrier shares and fp64 rate
                from the committed PMC passes of the same kernel - printed only while the plan of this run is the
                profiled one).
  cache_exceeding   BASELINE.json's 8.2M-tet beam on ONE GPU (exceeds the Infinity Cache; fused kernel, split stepping on
                three streams, with the one-launch-per-step figure beside it).
  per_gpu_of_8      the per-GPU workload of configs[3] / configs[4]: rank 3 of the 8 x-slabs of that beam, stepped plain,
                through the peer exchange with loop-back neighbours, through saa_step_synced with a one-rank RCCL
                communicator (eager launches / replayed graphs) and through sync-avoiding windows (native predictor at
                9126 inputs + 3000 predicted steps); `projected_8gpu` = what those step times would give on 8 GPUs if
                the exchange between GPUs cost what it costs inside one.
  unstructured  the same 1M-tet box meshed by a Delaunay triangulation of random points (no lattice anywhere).
  cpu_baseline  the CPU oracle ("port": SciPy CSR K.dot + the NumPy update, the reference's own per-step operations)
                timed on the SAME mesh on one host core and on min(8, cores) cores (one process per x-slab); the GPU
                steps the same mesh for the parity figure (parity.rel_l2).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# before the first HIP call of the process (synchronization_avoiding_algorithms_amd/hip_graphs.py: captured graphs with
# multi-block reductions replay wrongly from pre-recorded packets on this ROCm)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

E, NU, RHO, FZ, ALPHA, GAMMA = 1e6, 0.3, 1.0, 0.5, 0.5, 0.9
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
# fp64 vector peak: half the FP32 vector rate of MI355X_MICROARCH.md (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 flop/clk x
# 2.4 GHz; v_fma_f64 issues at half that rate) - AMD's datasheet figure for MI355X FP64 vector
FP64_VECTOR_PEAK = 78.6e12
# mesh refinement per GPU count: 150*n^3 tets ~ N * 1.03M
N_FOR_GPUS = {1: 19, 2: 24, 3: 27, 4: 30, 5: 32, 6: 34, 7: 36, 8: 38}
PROFILE_ROUNDS = ("r04", "r03")  # committed PMC summaries, newest first: profiles/<round>_pmc_summary.json / _onchip_summary.json
PLAN_IDENTITY = ("n_blocks", "max_owned", "max_local", "n_elem_copies", "n_halo_total", "lds_bytes", "threads", "n_items",
                 "n_pairs", "n_by_construction")  # what makes two plans the same work for the same kernel


def test_hook(point):
    """Tests only: ``SAA_BENCH_HOOKS=<module>`` (tests/bench_hooks.py) lets a test misbehave at a named point of the run
    (a preflight child that hangs, a collective that never returns); nothing of that lives in this file."""
    if os.environ.get("SAA_BENCH_HOOKS") == "tests.bench_hooks":  # (that one module, nothing else)
        from tests import bench_hooks

        bench_hooks.hook(point)


# ---------------------------------------------------------------------------------------------------------------
# deadline, legs, watchdog
# ---------------------------------------------------------------------------------------------------------------
def job_start_time():
    """Epoch seconds at which this JOB started: the start of the oldest ancestor process that is still part of it
    (``python bench.py --gpus N`` -> ``torch.distributed.run`` -> this rank), so that all ranks count the budget from
    the same instant and a launcher's slow first ``import torch`` on a fresh box is inside the budget, not in front of it."""
    env = os.environ.get("SAA_BENCH_T0")
    if env:
        return float(env)
    now = time.time()

    def part_of_job(argv):
        # `python bench.py ...` or `python -m torch.distributed.run ... bench.py ...` - not a shell, a test runner or a
        # profiler that merely carries "bench.py" somewhere in its command line
        if not argv or not os.path.basename(argv[0]).startswith("python"):
            return False
        rest = argv[1:]
        if len(rest) >= 2 and rest[0] == "-m":
            return rest[1] == "torch.distributed.run" and any(os.path.basename(a) == "bench.py" for a in rest[2:])
        return bool(rest) and os.path.basename(rest[0]) == "bench.py"

    try:
        with open("/proc/uptime") as fh:
            boot = now - float(fh.read().split()[0])
        tick = os.sysconf("SC_CLK_TCK")
        pid, t0 = os.getpid(), now
        for _ in range(4):
            with open(f"/proc/{pid}/stat") as fh:
                f = fh.read().rsplit(")", 1)[1].split()
            with open(f"/proc/{pid}/cmdline", "rb") as fh:
                argv = [a.decode(errors="replace") for a in fh.read().split(b"\0") if a]
            if pid != os.getpid() and not part_of_job(argv):
                break
            t0 = min(t0, boot + int(f[19]) / tick)  # field 22: start time in clock ticks since boot
            pid = int(f[1])                          # field 4: parent
            if pid <= 1:
                break
        return t0
    except (OSError, ValueError, IndexError):
        return now


class Legs:
    """Book-keeping of the run's legs and the watchdog that guarantees the line.  ``line`` is what rank 0 prints."""

    def __init__(self, t0, budget_s, rank):
        self.t0, self.budget, self.rank = t0, budget_s, rank
        self.line = None                    # the JSON object, once the headline exists
        self.status, self.seconds = {}, {}
        self.lock = threading.Lock()
        self._cut = None                    # (deadline, leg) of the leg currently running under a cut-off
        self._done = threading.Event()
        self._started = {}
        threading.Thread(target=self._watch, daemon=True).start()

    def left(self):
        return self.t0 + self.budget - time.time()

    def begin(self, leg, limit_s=None):
        """Start of a leg; ``limit_s``: the watchdog cuts the run off (line printed, exit 3) if the leg lasts longer."""
        with self.lock:
            self._started[leg] = time.time()
            self.status[leg] = "running"
            self._cut = (time.time() + limit_s, leg) if limit_s is not None else None

    def end(self, leg, status="done"):
        with self.lock:
            self.seconds[leg] = round(time.time() - self._started.get(leg, time.time()), 3)
            self.status[leg] = status
            self._cut = None

    def skip(self, leg, why):
        with self.lock:
            self.status[leg] = f"skipped: {why}"

    def emit(self, final):
        """Print the line (rank 0).  Called by the main thread at the end, or by the watchdog when time is up."""
        with self.lock:
            if self.line is not None and self.rank == 0:
                out = dict(self.line)
                out["legs"] = dict(self.status)
                out["leg_seconds"] = dict(self.seconds, total=round(time.time() - self.t0, 3))
                out["budget_s"] = self.budget
                print(json.dumps(out), flush=True)
            if final:
                self._done.set()

    def _watch(self):
        margin = 8.0  # seconds before the deadline at which the line is printed
        while not self._done.wait(0.5):
            now = time.time()
            with self.lock:
                cut = self._cut
            why = None
            if cut is not None and now > cut[0]:
                why = f"timed out ({cut[1]})"
                with self.lock:
                    self.status[cut[1]] = "unfinished: leg time limit"
            elif now > self.t0 + self.budget - margin:
                why = "deadline"
                with self.lock:
                    for leg, st in self.status.items():
                        if st == "running":
                            self.status[leg] = "unfinished: deadline"
            if why is None:
                continue
            if self.line is None and self.rank == 0:
                sys.stderr.write(f"bench: {why} before the headline was measured; legs: {self.status}\n")
            self.emit(final=False)
            sys.stderr.flush()
            if self.rank != 0:
                time.sleep(1.5)  # rank 0's line first
            os._exit(3)  # non-zero: a hang or an overrun must be seen (the line above is still valid)


# ---------------------------------------------------------------------------------------------------------------
def bench_mesh(n, kind):
    """The bench's mesh: the structured ``25n x n x n`` Kuhn-tet beam, or (``--mesh jittered``) the same beam with every
    interior node moved by up to 20 % of the cube edge and nodes and elements renumbered at random, or
    (``--mesh delaunay``) the same box with the same number of nodes meshed by a Delaunay triangulation of random points -
    no lattice anywhere, the class of mesh the reference's own Gmsh input is (Mesh_info/beam_US.geo:2-16,
    Mesh_info/beam_coarse.vtk)."""
    from synchronization_avoiding_algorithms_amd.mesh import Mesh, structured_beam

    if kind == "delaunay":
        from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam

        return delaunay_beam(n)
    mesh = structured_beam(n)
    if kind == "structured":
        return mesh
    rng = np.random.default_rng(0)
    pts = mesh.points.copy()
    h = 1.0 / n
    inner = np.all((pts > 1e-9) & (pts < np.array([25.0, 1.0, 1.0]) - 1e-9), axis=1)
    pts[inner] += rng.uniform(-0.2 * h, 0.2 * h, size=(int(inner.sum()), 3))
    perm = rng.permutation(len(pts))          # new id of old node i
    new_pts = np.empty_like(pts)
    new_pts[perm] = pts
    tets = perm[mesh.tets][rng.permutation(len(mesh.tets))]
    return Mesh(new_pts, {"tetra": tets, "triangle": perm[mesh.triangles]})


def build_rank_solver(mesh, n_parts, rank, device, block_nodes=0, threads=0):
    """One rank's solver outside a process group (tools, the parity leg): layout of that rank only, set-up fields from
    the HIP kernels; dt from this rank's elements and their neighbours (on the uniform synthetic beams that is the global
    CFL step; inside a process group PartitionedSolver takes the minimum over the ranks)."""
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import slab_partition

    lmd, mu = fs.lame(E, NU)
    epart = slab_partition(mesh, n_parts) if n_parts > 1 else np.zeros(len(mesh.tets), dtype=np.int64)
    lay, gshared, l_M, F_rankwise, dt = fs.rank_problem(mesh.points, mesh.tets, None, epart, rank, n_parts,
                                                         E, NU, RHO, FZ, GAMMA, device, facets=mesh.triangles)
    sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, l_M, F_rankwise, lay.dirichlet_dofs, lmd, mu,
                                dt, ALPHA, shared_local=lay.shared_local, shared_slots=lay.shared_slots,
                                n_global_shared=len(gshared), device=device, block_nodes=block_nodes, threads=threads)
    return sol, lay, gshared, dt


def measured_copy_bandwidth(device=0, n_bytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + written bytes per second) of this GPU, by the library's own 16-byte-per-lane
    copy kernel (``saa_device_copy_bandwidth``; MI355X_MICROARCH.md quotes 6.29 TB/s for such a copy): the practical
    HBM ceiling SURVEY.md section 8(d) asks to report next to the nominal 8 TB/s."""
    import ctypes as C

    from synchronization_avoiding_algorithms_amd import _lib

    bw = C.c_double()
    _lib.check(_lib.load().saa_device_copy_bandwidth(int(device), int(n_bytes), int(reps), C.byref(bw)))
    return bw.value


def cpu_baseline_and_parity(n, one_core_steps=300, mp_steps=1000, legs=None):
    """Oracle (CPU port of the reference's per-step operations: SciPy CSR ``K.dot`` + the NumPy update expression) timed
    on the bench's own mesh: on ONE host core (SciPy's SpMV is single-threaded) and on P = min(8, cores) cores the way
    the reference runs distributed (one process per x-slab, shared-node forces summed every step:
    ``oracle/cpu_baseline_mp.py``).  The matrix is assembled like the reference assembles it (element matrices added
    in element order, ``fem_oracle.assemble_local_stiffness_blocked``).  The GPU steps the same mesh from the same
    state for the parity figure.  With little time left (``legs.left()``) the step counts shrink, and below ~100 s the
    150k-tet beam stands in - the line says which mesh was timed and why."""
    from oracle import cpu_baseline_mp
    from oracle import fem_oracle as fo
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    left = legs.left() if legs is not None else 1e9
    fallback = None
    if left < 100.0 and n > 10:  # (measured on the build host: 25 s of set-up + 45 ms per step at 1M tets on one core)
        fallback = f"only {left:.0f} s of the budget were left: the 1M-tet set-up (~25 s) + steps did not fit"
        n, one_core_steps, mp_steps = 10, 3000, 15000
    mesh = structured_beam(n)
    t_setup = time.perf_counter()
    ranks, dt, _, _ = fo.setup_problem(mesh.points, mesh.tets, mesh.triangles, 1, np.zeros(len(mesh.tets), dtype=int))
    t_setup = time.perf_counter() - t_setup
    rp = ranks[0]
    d0 = np.zeros((len(rp.local_dof), 1))
    dn = np.zeros_like(d0)
    tn = 0
    t0 = time.perf_counter()
    for _ in range(one_core_steps):
        d1 = fo.explicit_step(rp.K, rp.F, rp.dirichlet, tn, dt, d0, dn, rp.l_M, ALPHA)
        dn, d0 = d0, d1
        tn = tn + dt
    cpu_s = time.perf_counter() - t0
    sol, lay, _, gdt = build_rank_solver(mesh, 1, 0, 0)
    assert gdt == dt and np.array_equal(lay.nodes, rp.nodes)
    sol.step(one_core_steps)
    g0, _, _ = sol.get_state()
    sol.close()
    rel = float(np.linalg.norm(g0 - d0) / np.linalg.norm(d0))
    ne = len(mesh.tets)
    what = f"synthetic beam n={n} ({ne} tets, {len(mesh.points)} nodes)"
    one = {"value": ne * one_core_steps / cpu_s, "unit": "element-updates/s", "cores": 1,
           "ms_per_step": 1e3 * cpu_s / one_core_steps,
           "sample": f"{what}, {one_core_steps} steps, scipy CSR K.dot (nnz {rp.K.nnz}) + numpy update "
                     f"(oracle/fem_oracle.py), {cpu_s:.2f} s after {t_setup:.1f} s of assembly"}
    cores = min(8, os.cpu_count() or 1)
    base = dict(one, kind="port")
    if cores > 1:
        try:
            mp = cpu_baseline_mp.run(cores, n, mp_steps)
            base = {"value": ne * mp_steps / mp["seconds"], "unit": "element-updates/s", "cores": cores, "kind": "port",
                    "ms_per_step": 1e3 * mp["seconds"] / mp_steps,
                    "sample": f"{what} in {cores} x-slabs, one process per slab ({os.cpu_count()} host cores), {mp_steps} "
                              f"steps, per step scipy CSR K.dot + sum of the {mp['n_shared']} shared nodes' forces "
                              f"over the ranks in rank order (shared memory) + numpy update "
                              f"(oracle/cpu_baseline_mp.py), {mp['seconds']:.2f} s",
                    "one_core": one}
        except Exception as e:  # noqa: BLE001 - the one-core figure is still a valid baseline
            base["multi_process_failed"] = str(e)[:200]
    if fallback:
        base["fallback"] = fallback
    return (base, {"rel_l2": rel, "mesh": what, "steps": one_core_steps, "tolerance": 1e-10,
                   "against": "oracle (assembled CSR of the same mesh), from rest under the ramped load"})


def preflight_main(world, rank, local_rank):
    """Child process of one rank (bench.py --preflight): a tiny partitioned problem stepped through the peer exchange
    on the real devices.  Exit code 0 = the direct xGMI path works here; anything else (including a crash of this
    process) makes the parent fall back to the RCCL all-reduce."""
    test_hook("preflight")
    import torch
    import torch.distributed as dist

    import datetime

    torch.cuda.set_device(local_rank)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))  # a crashed peer must not park the others
    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver
    from synchronization_avoiding_algorithms_amd.mesh import slab_partition, structured_beam

    mesh = structured_beam(4)
    part = PartitionedSolver(mesh.points, mesh.tets, mesh.triangles, slab_partition(mesh, world), rank, world, E=E, nu=NU,
                             rho=RHO, fz=FZ, alpha=ALPHA, gamma=GAMMA, device=local_rank, exchange="peer", wait_timeout_s=20)
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


def run_preflight(args, world, limit_s):
    """Runs preflight_main in a child process BEFORE this process touches the GPU; True iff it exited cleanly within
    ``limit_s`` seconds (the child and whatever it started are killed otherwise)."""
    import signal
    import subprocess

    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # the children's rank 0 hosts their own rendezvous store
    cmd = [sys.executable, os.path.abspath(__file__), "--preflight", "--gpus", str(world)]
    if args.same_device:
        cmd.append("--same-device")
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    except OSError:
        return False
    try:
        return proc.wait(timeout=max(1.0, limit_s)) == 0
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        return False


def sync_avoiding_leg(part, args, rank, world, ne_total, fence, train_seconds):
    """BASELINE.json configs[4] the way the reference's workflow produces it (README.md:33-38), all on the GPUs:
      1. ground truth  - the synchronised run from rest (Data_prepare.py:223-240), recording every rank's shared-dof
                         history (what Shared_extraction.py slices out) and the full state at the end of every window;
      2. training      - one LSTM per rank on ITS history (Model_training.py; training.train_on_history: windowing,
                         [-1, 0] scaling, Adam with 0.998^epoch decay, HIP-graph optimiser step): `--sa-train-epochs`
                         epochs (0: the reference's schedule, until the rate reaches lr_min), at most `train_seconds`;
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
    # 2. one model per rank, trained on its own history.  The epoch count is what decides the model (fixed seeds); the
    #    time bound only protects the run's deadline.  Ranks sharing one GPU (rehearsals) train one after the other:
    #    graph replays of two processes on one device get in each other's way.
    epochs = None
    if args.sa_train_epochs is not None:
        epochs = args.sa_train_epochs if args.sa_train_epochs > 0 else None  # 0: Model_training.py:65
        schedule = "fixed epoch count" if args.sa_train_epochs > 0 else "reference schedule (until lr_min)"
    else:  # the reference's schedule as far as the time allowed goes (3450 epochs: 80 s at 9126 inputs on the fused training path)
        epochs, schedule = None, "reference schedule (until lr_min), cut off by the time allowed if that comes first"
    t0 = time.perf_counter()
    turns = world if args.same_device else 1
    for turn in range(turns):
        if turns == 1 or turn == rank:
            with torch.enable_grad():
                model, smax, smin, tl, vl = tr.train_on_history(truth, n_s, n_p, n_f, cut_off=1.0, seed=1234 + rank,
                                                                hidden_size=hid, num_epochs=epochs,
                                                                max_seconds=train_seconds / turns,
                                                                verbose=True, rank=0 if turns > 1 else rank,
                                                                log=sys.stderr)  # (progress: stdout is the line's)
        if turns > 1:
            fence()
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
        for _ in range(3):  # untimed (the PyTorch-ROCm route would capture its HIP graph on the third call)
            predictor(start, hist)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            predictor(start, hist)
        ev[1].record()
        torch.cuda.synchronize()
        predictor_ms = ev[0].elapsed_time(ev[1]) / 5
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
    ep = torch.tensor([len(tl)], dtype=torch.int64, device="cuda")
    dist.all_reduce(ep, op=dist.ReduceOp.MIN)
    finite = torch.tensor([float(torch.isfinite(hist[-1]).all())], device="cuda")
    dist.all_reduce(finite, op=dist.ReduceOp.MIN)
    return {"value": ne_total * windows * win / elapsed, "unit": "element-updates/s",
            "ms_per_step": 1e3 * elapsed / (windows * win), "steps": windows * win, "windows": windows,
            "n_past": n_p, "n_future": n_f, "filter_size": n_s, "input_size_rank0": width,
            "synchronised_steps_before": start,
            "predictor": {"backend": predictor.backend, "ms_per_window_rank0": predictor_ms,
                          "note": "one call = the table of a whole window (all filter_size phases); inside the timed region"},
            "rel_l2_vs_synchronised": errs, "state_finite": bool(finite.item()),
            "training": {"truth_steps": n_truth, "windows": int(groups), "epochs": len(tl),
                         "epochs_min_over_ranks": int(ep.item()), "schedule": schedule, "seconds": train_s,
                         "seconds_allowed": train_seconds,
                         "train_mse_first_last": [tl[0], tl[-1]], "validation_mse_last": vl[-1], "hidden_size": hid},
            "note": f"after {start} synchronised steps every rank's LSTM (trained in this run on the synchronised "
                    f"history of its own shared dofs) predicts them for {win}-step windows; no collective inside a "
                    "window; predictor time included; rel_l2_vs_synchronised = whole displacement field against the "
                    "synchronised run at the end of each window"}


def predictor_leg(device_index):
    """configs[4]'s predictor at the width an interior rank of the 8-GPU run has (3042 shared nodes = 9126 inputs, H = 50,
    n_past = n_future = 20, filter 150: DNN_prediction.py:38-55 for one window), seeded random weights, synthetic history:
    the library's own kernels (saa_predictor_*) and, beside them, the PyTorch-ROCm route replayed as a HIP graph.  The
    dominant kernel is GEMM-shaped and runs on the f32 matrix cores, so its yardstick is the MFMA peak, not HBM."""
    import torch

    from synchronization_avoiding_algorithms_amd import predictor as pr

    I, H, n_p, n_f, n_s = 9126, 50, 20, 20, 150
    dev = torch.device("cuda", device_index)
    torch.manual_seed(1)
    model = pr.LSTM_encoder_decoder(I, H).to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(2)
    hist = torch.cumsum(torch.randn(n_p * n_s + 64, I, generator=gen, device=dev, dtype=torch.float64) * 1e-4, 0)
    smax, smin = float(hist.max()) * 1.05, float(hist.min()) * 1.05
    n = n_p * n_s + 17

    def ms_per_call(fn, reps):
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize(dev)
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize(dev)
        return ev[0].elapsed_time(ev[1]) / reps

    nat = pr.NativePredictor(model, n_p, n_f, n_s, device_index)
    table = nat.predict(n, hist, smax, smin).clone()
    t_nat = ms_per_call(lambda: nat.predict(n, hist, smax, smin, table), 40)
    with torch.no_grad():
        graph = pr.DevicePredictor(model, n_p, n_f, n_s, smax, smin, backend="torch")
        for _ in range(4):
            want = graph(n, hist)
        t_pt = ms_per_call(lambda: graph(n, hist), 10)
        diff = float((table - want).abs().max() / want.abs().max())
    nat.close()
    D, G, M = 2 * H, 8 * H, n_p * n_s
    flop = 2.0 * (M * I * G + n_s * I * G + n_f * n_s * D * I) + 2.0 * n_s * (n_p * (2 * H * G + D * G) + n_f * D * G)
    return {"workload": f"one prediction window: {n_s} phases, n_past = n_future = {n_p}, input_size {I} (an interior rank of "
                        "configs[4]), hidden 50, fp32 like the reference, seeded random weights, synthetic history",
            "backend": "native HIP (saa_predictor_*): 2 f32-MFMA GEMMs + recurrence kernel + output GEMM",
            "ms_per_window": t_nat, "pytorch_rocm_hip_graph_ms_per_window": t_pt, "speedup_vs_pytorch_rocm": t_pt / t_nat,
            "max_difference_vs_pytorch_rocm_over_range": diff,
            "flop_per_window": flop, "TFLOPs": flop / (t_nat * 1e-3) / 1e12,
            "roofline": {"bound": "mfma", "peak": 157.3, "unit": "TFLOP/s", "achieved": flop / (t_nat * 1e-3) / 1e12,
                         "frac": flop / (t_nat * 1e-3) / 1e12 / 157.3,
                         "note": "whole window (four launches) against the dense f32 matrix peak; the input-projection "
                                 "GEMM alone (21.9 of the 28.6 GFLOP): 77 TFLOP/s = 0.49 "
                                 "(profiles/r03_predictor_kernel_trace_saa.csv)"},
            "share_of_a_3000_step_window": f"{t_nat:.2f} ms against ~26 ms of exchange-free stepping per rank "
                                           f"(PyTorch-ROCm: {t_pt:.1f} ms)"}


def kernel_sources_digest():
    """Short digest of the step kernels' sources: a committed counter summary records the one it was measured with."""
    import hashlib

    h = hashlib.sha256()
    for name in ("saa_kernels.hip", "saa_device.h", "saa_plan.h"):
        with open(os.path.join(REPO, "synchronization_avoiding_algorithms_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_counters(key, stats):
    """HBM traffic per launch and the on-chip shares of the kernel `key` ("resident19", "fused38", ...) from the newest
    committed PMC summary (tools/pmc_collect.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes,
    nothing else enabled; KiB units; FETCH_SIZE doubled: gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md "HBM").
    They are figures of ANOTHER run of the same command, so they are only handed out while the plan of this run
    (`stats`) is the plan that was profiled; otherwise the reason is returned instead.
    Returns (traffic_bytes | None, onchip record | None, source | None, note | None)."""
    for rnd in PROFILE_ROUNDS:
        try:
            with open(os.path.join(REPO, "profiles", f"{rnd}_pmc_summary.json")) as fh:
                pmc = json.load(fh)
            fetch = pmc[f"{rnd}_{key}:FETCH_SIZE"]["mean_per_dispatch"]
            write = pmc[f"{rnd}_{key}:WRITE_SIZE"]["mean_per_dispatch"]
        except (OSError, KeyError, ValueError):
            continue
        recorded = pmc.get(f"{rnd}_{key}:plan")
        if recorded is None:
            return None, None, None, (f"profiles/{rnd}_pmc_summary.json holds counters for {key} but not the plan they were "
                                      "measured on: not printed")
        diff = {k: (recorded.get(k), stats.get(k)) for k in PLAN_IDENTITY if recorded.get(k) != stats.get(k)}
        if diff:
            return None, None, None, (f"the plan of this run differs from the one profiled in profiles/{rnd}_pmc_summary.json "
                                      f"(recorded, now): {diff} - counters not printed, profile again")
        note = None
        if recorded.get("kernel_sources") not in (None, kernel_sources_digest()):
            note = "the kernel sources changed since the counters were collected (same plan)"
        onchip = None
        try:
            with open(os.path.join(REPO, "profiles", f"{rnd}_onchip_summary.json")) as fh:
                onchip = json.load(fh)[key]
        except (OSError, KeyError, ValueError):
            pass
        src = f"profiles/{rnd}_pmc_summary.json [{rnd}_{key}] (rocprofv3 --pmc, separate passes, launches of the same length)"
        return (2.0 * fetch + write) * 1024.0, onchip, src, note
    return None, None, None, None


def timed_steps(call, steps_per_call, min_ms, sync):
    """``call(k)`` enqueues k steps.  One settled call sizes the number of repetitions so that the timed region lasts
    >= min_ms; returns (seconds per step, calls)."""
    call(steps_per_call)
    sync()
    t0 = time.perf_counter()
    call(steps_per_call)
    sync()
    est = max(time.perf_counter() - t0, 1e-7)
    calls = int(min(max(1, np.ceil(min_ms * 1e-3 / est)), 100000))
    for attempt in range(3):
        t0 = time.perf_counter()
        for _ in range(calls):
            call(steps_per_call)
        sync()
        total = time.perf_counter() - t0
        if total >= min_ms * 1e-3 or attempt == 2:
            break
        calls = int(min(np.ceil(1.15 * calls * min_ms * 1e-3 / max(total, 1e-7)), 100000))
    return total / (calls * steps_per_call), calls


def contract_roofline(ne, nn, s_per_step):
    """algorithmic bytes per step 16*Ne + 216*Nn against the nominal HBM peak."""
    b_alg = 16 * ne + 216 * nn
    return {"algorithmic_bytes_per_step": b_alg, "achieved_GBps": b_alg / s_per_step / 1e9,
            "frac": b_alg / s_per_step / HBM_PEAK}


def cache_exceeding_leg(mesh38, device, min_ms):
    """BASELINE.json's 8.2M-tet beam on ONE GPU (the point that exceeds the Infinity Cache): 2048
    blocks, two per CU, one launch of the fused kernel per step."""
    t0 = time.perf_counter()
    sol, lay, _, dt = build_rank_solver(mesh38, 1, 0, device)
    setup_s = time.perf_counter() - t0
    ne, nn = len(lay.cells_local), len(lay.nodes)
    res = sol.resident_kernel_info()
    sol.time_steps(300)                                   # settle the clocks on this path
    est = sol.time_steps(200) / 200.0                     # ms per step
    short_steps = 1000
    short = sol.time_steps(short_steps) / short_steps
    n = int(min(max(200, np.ceil(min_ms / max(est, 1e-6))), 200000))
    ms = sol.time_steps(n)
    s_step = ms * 1e-3 / n
    # the same steps with ONE launch of all blocks per step (what rounds 1-3 measured): plans of four rounds of workgroups
    # and more step their blocks as three sets on three streams by default (saa_set_option("split_stepping"))
    sol.set_option("split_stepping", 0)
    sol.time_steps(200)
    plain_ms = sol.time_steps(max(n // 3, 200)) / max(n // 3, 200)
    sol.set_option("split_stepping", 1)
    stats = sol.plan_stats()
    sol.synchronize()
    sol.close()
    key = ("resident" if res["capable"] else "fused") + "38"
    out = {"workload": f"synthetic 25n x n x n Kuhn-tet beam: {ne} tets, {nn} nodes, ONE partition on one GPU, fp64, dt={dt:.6e}",
           "kernel": "persistent_steps_kernel<false,false>" if res["capable"] else "fused_step_kernel<false> (one launch per step)",
           "ms_per_step": 1e3 * s_step, "steps_timed": n, "element_updates_per_s": ne / s_step,
           "short_region": {"ms_per_step": short, "steps": short_steps},
           "schedule": ("split stepping: the blocks as left / middle / right sets on three streams tied by events, launch "
                        "boundaries hidden under the other sets' work" if stats["n_blocks"] >= 2048 and not res["capable"]
                        else "one launch of all blocks per step"),
           "one_launch_per_step_ms": plain_ms, "setup_seconds": setup_s, "plan": stats}
    out["roofline"] = contract_roofline(ne, nn, s_step)
    traffic, onchip, src, note = committed_counters(key, stats)
    if traffic is not None:
        out["roofline"].update(traffic=traffic, traffic_source=src,
                               hbm_measured={"GBps": traffic / s_step / 1e9, "frac_of_peak": traffic / s_step / HBM_PEAK,
                                             "traffic_over_algorithmic": traffic / out["roofline"]["algorithmic_bytes_per_step"]})
    if onchip is not None:
        out["roofline"]["onchip"] = {k: onchip[k] for k in ("valu_busy", "lds_busy", "lds_bank_conflict_share",
                                                          "wave_wait_share", "wave_issue_stall_share") if k in onchip}
    if note:
        out["roofline"]["counters_note"] = note
    return out


def per_gpu_of_8_leg(mesh38, device, min_ms):
    """What ONE GPU of BASELINE.json's configs[3] / configs[4] does per step: rank 3 of the 8 x-slabs of the 8.2M-tet beam
    (an interior slab: two interfaces, 3042 shared nodes), on one GPU, through every route a step can take -
      plain          exchange-free resident steps (what a predicted window runs between predictor calls);
      peer_loopback  saa_step_peer with loop-back neighbours (saa_peer_attach_loopback: push, stamped entries, rank-ordered
                     sum all run; the delivery is local memory instead of xGMI);
      rccl           saa_step_synced with a ONE-rank RCCL communicator: fused kernel -> ncclAllReduce -> finish kernel,
                     enqueued one by one (eager) and as replayed HIP graphs of three steps (graph);
      sync_avoiding  windows of 3000 predicted steps, each preceded by the native predictor call at 9126 inputs (seeded
                     random weights: the cost does not depend on them) - Online_predictor.py:277-318.
    `projected_8gpu`: all tets of the beam (8 230 800) / that step time, i.e. what 8 GPUs would deliver if the exchange between them cost
    what it costs inside one (no xGMI latency, no skew between ranks)."""
    import torch

    from synchronization_avoiding_algorithms_amd import predictor as pr

    world, rank, ne_total = 8, 3, len(mesh38.tets)
    dev = torch.device("cuda", device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    t0 = time.perf_counter()
    sol, lay, gshared, dt = build_rank_solver(mesh38, world, rank, device)
    setup_s = time.perf_counter() - t0
    sol.set_stream(stream)
    ne, nn, nsh = len(lay.cells_local), len(lay.nodes), len(lay.shared_local)
    res = sol.resident_kernel_info()
    spl = res["steps_per_launch"] if res["capable"] else 100
    out = {"workload": f"rank {rank} of {world} x-slabs of the synthetic 25n x n x n beam of {ne_total} tets: {ne} tets, {nn} nodes, "
                       f"{nsh} shared nodes ({3 * nsh} predicted dofs), all-reduce buffer {3 * len(gshared)} doubles",
           "resident": bool(res["capable"]), "setup_seconds": setup_s, "plan": sol.plan_stats(), "routes": {}}

    def route(name, s_step, **extra):
        out["routes"][name] = dict({"us_per_step": 1e6 * s_step, "element_updates_per_s_this_gpu": ne / s_step,
                                    "contract_frac": contract_roofline(ne, nn, s_step)["frac"]}, **extra)

    sol.step(4 * spl)  # clocks settle over the first tens of milliseconds of load
    s, calls = timed_steps(sol.step, spl, min_ms, sol.synchronize)
    route("plain", s, calls=calls, steps_per_call=spl)
    sol.peer_attach_loopback(2)
    s, calls = timed_steps(sol.step_peer, spl, min_ms, sol.synchronize)
    route("peer_loopback", s, calls=calls, steps_per_call=spl,
          note="every shared node has one imaginary co-holder living in this rank's own inbox")
    sol.close()

    sol, lay, gshared, dt = build_rank_solver(mesh38, world, rank, device)
    sol.set_stream(stream)
    iface = torch.zeros(3 * len(gshared), dtype=torch.float64, device=dev)
    sol.set_interface_buffer(iface)
    sol.comm_init(sol.comm_unique_id(), 0, 1)
    for name, flag in (("rccl_eager", 0), ("rccl_graph", 1)):
        sol.set_option("synced_graph", flag)
        s, calls = timed_steps(sol.step_synced, 300, min_ms, sol.synchronize)
        route(name, s, calls=calls, steps_per_call=300,
              note="one-rank communicator: the collective is as cheap as it gets, what is measured is the launch route")
    # sync-avoiding windows: predictor call + 3000 predicted steps, history recorded by the step kernels
    n_p, n_f, n_s, hid = 20, 20, 150, 50
    warm, win, width = n_p * n_s, n_f * n_s, 3 * nsh
    windows = int(min(max(3, np.ceil(min_ms * 1e-3 / (win * out["routes"]["plain"]["us_per_step"] * 1e-6))), 40))
    torch.manual_seed(1)
    model = pr.LSTM_encoder_decoder(width, hid).to(dev).eval()
    hist = torch.zeros((warm + windows * win, width), dtype=torch.float64, device=dev)
    zero = np.zeros(sol.n_dof)
    sol.set_state(zero, zero, 0.0)
    sol.set_option("synced_graph", 0)
    sol.step_synced(warm, hist, 0)            # the synchronised warm-up records the history the first prediction reads
    sol.synchronize()
    a = float(hist[:warm].abs().max()) * 1.05 + 1e-30
    nat = pr.NativePredictor(model, n_p, n_f, n_s, device)
    table = nat.predict(warm, hist, a, -a).clone()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(dev)
    ev[0].record()
    for _ in range(10):
        nat.predict(warm, hist, a, -a, table)
    ev[1].record()
    torch.cuda.synchronize(dev)
    pred_ms = ev[0].elapsed_time(ev[1]) / 10
    i = warm
    sol.synchronize()
    t0 = time.perf_counter()
    for _ in range(windows):
        nat.predict(i, hist, a, -a, table)
        sol.step_predicted(win, table, 0, hist, i)
        i += win
    sol.synchronize()
    torch.cuda.synchronize(dev)
    s = (time.perf_counter() - t0) / (windows * win)
    finite = bool(torch.isfinite(hist[-1]).all().item())
    route("sync_avoiding", s, windows=windows, steps_per_window=win, predictor_ms_per_window=pred_ms,
          predictor_share=pred_ms * 1e-3 / (s * win), state_finite=finite,
          note="per window: saa_predictor_predict (four launches) + saa_step_predicted(3000); no exchange of any kind")
    nat.close()
    sol.close()
    out["projected_8gpu"] = {
        "element_updates_per_s": {k: ne_total / (v["us_per_step"] * 1e-6) for k, v in out["routes"].items() if k != "plain"},
        "assumptions": "all 8 ranks step like this interior slab (the two end slabs have one interface and are not slower: "
                       "profiles/r03_rank_survey.txt); the exchange between GPUs costs what it costs inside one - local "
                       "memory latency instead of xGMI's (tools/peer_latency.py: +0.2 us per step at 1 us delivery time, "
                       "+0.5 at 2 us), a one-rank all-reduce instead of an 8-rank ring; no skew between ranks.  A "
                       "projection from one-GPU measurements, NOT a measurement on 8 GPUs"}
    return out


def unstructured_leg(n, device, min_ms):
    """The 1M-tet box meshed WITHOUT any lattice: a Delaunay triangulation of random points (mesh.delaunay_beam), the
    class of mesh the reference's own Gmsh input belongs to (Mesh_info/beam_US.geo:2-16).  Same material, load, clamp."""
    t0 = time.perf_counter()
    mesh = bench_mesh(n, "delaunay")
    mesh_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    sol, lay, _, dt = build_rank_solver(mesh, 1, 0, device)
    setup_s = time.perf_counter() - t0
    ne, nn = len(mesh.tets), len(mesh.points)
    res = sol.resident_kernel_info()
    spl = res["steps_per_launch"] if res["capable"] else 100
    sol.time_steps(4 * spl)
    est = sol.time_steps(2 * spl) / (2 * spl)
    n_steps = int(min(max(2 * spl, np.ceil(min_ms / max(est, 1e-6) / spl) * spl), 2000000))
    s_step = sol.time_steps(n_steps) * 1e-3 / n_steps
    stats = sol.plan_stats()
    sol.synchronize()
    sol.close()
    out = {"workload": f"Delaunay triangulation of {nn} random points in the 25 x 1 x 1 box (boundary points on a grid of "
                       f"the mean spacing): {ne} tets, one partition, fp64, dt={dt:.6e}",
           "kernel": "persistent_steps_kernel<false,false>" if res["capable"] else "fused_step_kernel<false>",
           "ms_per_step": 1e3 * s_step, "steps_timed": n_steps, "element_updates_per_s": ne / s_step,
           "roofline": contract_roofline(ne, nn, s_step), "lds_conflict_factor": stats["lds_conflict_factor"],
           "lds_atomic_conflict_factor": stats["lds_atomic_conflict_factor"], "plan": stats,
           "mesh_seconds": mesh_s, "setup_seconds": setup_s}
    traffic, onchip, src, note = committed_counters(("resident" if res["capable"] else "fused") + f"{n}_delaunay", stats)
    if traffic is not None:  # (per launch of `spl` steps, like the headline's)
        out["roofline"].update(traffic=traffic, traffic_source=src,
                               hbm_measured={"GBps": traffic / (s_step * spl) / 1e9, "frac_of_peak": traffic / (s_step * spl) / HBM_PEAK,
                                             "traffic_over_algorithmic": traffic / (spl * out["roofline"]["algorithmic_bytes_per_step"])})
    if onchip is not None:
        out["roofline"]["onchip"] = {k: onchip[k] for k in ("valu_busy", "lds_busy", "lds_bank_conflict_share",
                                                          "wave_wait_share", "wave_issue_stall_share") if k in onchip}
    if note:
        out["roofline"]["counters_note"] = note
    return out


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks (one process per GPU) through
    ``torch.distributed.run`` - the reference's whole launch story is ``mpirun -np P python3 ...``
    (/root/reference README.md:33-38) - BEFORE this process makes any GPU call (it never does), relay rank 0's single
    JSON line and exit non-zero if the ranks failed or printed no line.  The ranks keep the deadline themselves; this
    parent ends their process group if they are still there 30 s after it."""
    import signal
    import socket
    import subprocess

    t0 = job_start_time()
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env["SAA_BENCH_T0"] = repr(t0)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the peer exchange need it on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)

    def reaper():
        if proc.poll() is None:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass

    timer = threading.Timer(max(5.0, t0 + args.budget_s + 30.0 - time.time()), reaper)
    timer.daemon = True
    timer.start()
    line = None
    try:
        for ln in proc.stdout:  # rank 0's JSON line is the only thing the ranks write to stdout
            try:
                if ln.lstrip().startswith("{") and "metric" in json.loads(ln):
                    line = ln.strip()
                    print(line, flush=True)  # at once: whatever happens to the ranks later, the line is out
                    continue
            except ValueError:
                pass
            sys.stderr.write(ln)
        rc = proc.wait()
    except BaseException:  # Ctrl-C: end exactly the process group started above
        reaper()
        raise
    timer.cancel()
    if rc != 0 or line is None:
        raise SystemExit(rc if rc != 0 else 4)


def single_gpu_legs(args, legs, put, sol, stats, n, ne_total, nn_total, device):
    """The extra legs of the N = 1 line, each inside the run's budget: roofline of the headline kernel, the per-GPU
    workload of configs[3] / [4], the cache-exceeding point, the unstructured mesh, the predictor, the CPU baseline."""
    wanted = {"roofline", "per_gpu_of_8", "cache_exceeding", "unstructured", "predictor", "cpu_baseline"}
    if args.legs not in ("all", ""):
        wanted = set() if args.legs == "none" else {w.strip() for w in args.legs.split(",")}
    if args.no_cpu_baseline:
        wanted.discard("cpu_baseline")

    def run(leg, need_s, fn, key=None):
        """One leg: skipped when not wanted or when less than `need_s` seconds of the budget are left; a failure is
        reported in the line and never costs the headline."""
        if leg not in wanted:
            legs.skip(leg, "not requested (--legs / --no-cpu-baseline)")
            return
        if legs.left() < need_s:
            legs.skip(leg, f"{legs.left():.0f} s of the budget left, needs ~{need_s:.0f}")
            return
        legs.begin(leg)
        try:
            value = fn()
            if key is not None:
                put(key, value)
            legs.end(leg)
        except Exception as exc:  # noqa: BLE001
            put(key or leg, {"error": repr(exc)[:500]})
            legs.end(leg, "failed")

    def roofline():
        # the dominant (only) kernel, HIP events on the kernel's own stream.  One launch of the resident kernel advances
        # `spl` steps (one launch = spl * Ne element-updates); without it (plan does not fit LDS / not all workgroups
        # co-resident) one launch of the fused kernel is one step
        res = sol.resident_kernel_info()
        spl = res["steps_per_launch"] if res["capable"] else 1
        sol.time_steps(2 * spl if spl > 1 else 200)  # settle the clocks on this very path
        short_launches = 12 if spl > 1 else 3000
        short_ms = sol.time_steps(short_launches * spl)
        est = short_ms / short_launches                                    # ms per launch
        launches = int(min(max(short_launches, np.ceil(max(args.min_timed_ms, 1000.0) / est)), 2000000 // spl))
        ms = sol.time_steps(launches * spl)
        b_alg = 16 * ne_total + 216 * nn_total
        dur_s = ms * 1e-3 / (launches * spl)  # per step
        achieved = b_alg / dur_s
        roof = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": None,
                "kernel": "persistent_steps_kernel<false,false>" if spl > 1 else "fused_step_kernel<false>",
                "steps_per_launch": spl, "launches_timed": launches, "avg_launch_us": 1e3 * ms / launches,
                "us_per_step": 1e6 * dur_s, "timed_region_ms": ms,
                "short_region": {"launches": short_launches, "avg_launch_us": 1e3 * short_ms / short_launches,
                                 "us_per_step": 1e3 * short_ms / (short_launches * spl),
                                 "frac": b_alg / (short_ms * 1e-3 / (short_launches * spl)) / HBM_PEAK,
                                 "note": "the ~90 ms region earlier rounds quoted; the figures of this object come from "
                                         "the >= 1 s region (the chip settles at a lower clock under sustained load)"},
                "algorithmic_bytes_per_step": b_alg, "algorithmic_bytes_per_launch": b_alg * spl,
                "algorithmic_bytes_per_element_update": b_alg / ne_total,
                "note": "achieved / frac are the CONTRACT figure: algorithmic bytes (SURVEY.md section 8(d): what a kernel "
                        "that re-reads the partition every step must move) per second against the HBM peak - an "
                        "equivalent bandwidth, not the kernel's HBM traffic.  What it moves is `traffic` / `hbm_measured`; "
                        "what bounds it is `bound` / `onchip`."}
        key = (f"resident{n}" if spl > 1 else f"fused{n}") + {"structured": "", "jittered": "_jittered",
                                                               "delaunay": "_delaunay"}[args.mesh]
        traffic, onchip, src, note = committed_counters(key, stats)
        if traffic is not None:
            roof["traffic"] = traffic
            roof["traffic_source"] = src
            hbm = traffic / (dur_s * spl)
            roof["hbm_measured"] = {"GBps": hbm / 1e9, "frac_of_peak": hbm / HBM_PEAK,
                                    "traffic_over_algorithmic": traffic / (b_alg * spl)}
        if note:
            roof["counters_note"] = note
        if onchip is not None:
            oc = {k: onchip[k] for k in ("valu_busy", "lds_busy", "lds_bank_conflict_share", "barrier_wait_share",
                                         "wave_wait_share", "wave_issue_stall_share") if k in onchip}
            if "fp64_flop_per_step" in onchip:  # executed fp64 flops (element copies included), counted by the SQ
                flops = onchip["fp64_flop_per_step"] / dur_s
                oc.update(fp64_flops_per_s=flops, frac_of_fp64_vector_peak=flops / FP64_VECTOR_PEAK,
                          fp64_vector_peak=FP64_VECTOR_PEAK, fp64_flop_per_step=onchip["fp64_flop_per_step"],
                          fp64_flop_per_element_update=onchip["fp64_flop_per_step"] / ne_total)
            oc["source"] = onchip.get("source")
            roof["onchip"] = oc
            if roof.get("hbm_measured", {}).get("frac_of_peak", 1.0) < 0.5:
                # the counters say what the limiter is: name it where a reader looks first
                roof["bound"] = "on-chip (fp64 vector issue + LDS pipe; the partition is LDS-resident, HBM is not the limiter)"
                roof["contract_bound"] = "hbm"
        copy_bw = measured_copy_bandwidth(device)
        roof["measured_copy_GBps"] = copy_bw / 1e9
        roof["measured_copy_kernel"] = "saa_device_copy_bandwidth: one 16-byte element per thread, 1 GiB -> 1 GiB, 10 launches"
        roof["frac_of_measured_copy"] = achieved / copy_bw
        return roof

    shared = {}

    def roofline_and_keep():
        roof = roofline()
        shared["structured_us"] = roof["us_per_step"]
        return roof

    run("roofline", 10.0, roofline_and_keep, "roofline")
    sol.close()
    mesh38 = {}

    def get38():
        if "m" not in mesh38:
            from synchronization_avoiding_algorithms_amd.mesh import structured_beam

            mesh38["m"] = structured_beam(args.big_refine)
        return mesh38["m"]

    if args.mesh == "structured" and not args.refine:
        # the other BASELINE configurations, as far as one GPU can carry them (the driver has no 8-GPU node every round)
        run("per_gpu_of_8", 75.0, lambda: per_gpu_of_8_leg(get38(), device, args.min_timed_ms), "per_gpu_of_8")
        run("cache_exceeding", 110.0, lambda: cache_exceeding_leg(get38(), device, args.min_timed_ms), "cache_exceeding")
        mesh38.clear()

        def unstructured():
            leg = unstructured_leg(n, device, args.min_timed_ms)
            if "structured_us" in shared:  # same run, same box, same kind of timing (HIP events over >= 1 s)
                leg["step_time_over_structured"] = 1e3 * leg["ms_per_step"] / shared["structured_us"]
            return leg

        run("unstructured", 120.0, unstructured, "unstructured")
    else:
        for leg in ("per_gpu_of_8", "cache_exceeding", "unstructured"):
            legs.skip(leg, "only with the default mesh (--mesh structured, no --refine)")
    run("predictor", 60.0, lambda: predictor_leg(device), "predictor")

    def cpu():
        base, parity = cpu_baseline_and_parity(n if args.mesh == "structured" else 10, legs=legs)
        put("parity", parity)
        return base

    run("cpu_baseline", 45.0, cpu, "cpu_baseline")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--refine", type=int, default=0, help="override mesh refinement n (25n x n x n cubes)")
    ap.add_argument("--mesh", default="structured", choices=["structured", "jittered", "delaunay"],
                    help="jittered: the same beam with moved nodes and random numbering; delaunay: the same box meshed by a "
                         "Delaunay triangulation of random points (an unstructured mesh without any lattice)")
    ap.add_argument("--block-nodes", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--budget-s", type=float, default=420.0,
                    help="deadline of the whole run in seconds from the start of the job; legs that do not fit are skipped, "
                         "a leg that overruns is cut off (the line is printed either way once the headline exists)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--no-sync-avoiding", action="store_true", help="N > 1: skip the sync-avoiding-mode leg")
    ap.add_argument("--sa-windows", type=int, default=3, help="sync-avoiding leg: predicted windows that are timed")
    ap.add_argument("--sa-filter", type=int, default=150, help="sync-avoiding leg: filter_size n_s (Online_predictor.py:59)")
    ap.add_argument("--sa-truth-steps", type=int, default=30000,
                    help="sync-avoiding leg: synchronised steps recorded as training data")
    ap.add_argument("--sa-train-epochs", type=int, default=None,
                    help="sync-avoiding leg: epochs of training per rank (0 = the reference's schedule, "
                         "Model_training.py:65); default: the reference's schedule, cut off at --sa-train-seconds")
    ap.add_argument("--sa-train-seconds", type=float, default=120.0,
                    help="sync-avoiding leg: cap on the training time per rank (further capped by the run's budget)")
    ap.add_argument("--no-rccl-leg", action="store_true",
                    help="N > 1: skip the extra measurement with the RCCL all-reduce when the peer exchange is in use")
    ap.add_argument("--torch-exchange", action="store_true",
                    help="N > 1: all-reduce through torch.distributed (same as --exchange torch)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer", "rccl", "torch"],
                    help="N > 1: how shared-node forces travel: peer = direct xGMI stores (saa_step_peer), rccl = "
                         "ncclAllReduce from C++, torch = torch.distributed.all_reduce; auto tries them in that order")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-split-stepping", action="store_true",
                    help="plans of four rounds of workgroups and more: one launch of all blocks per step instead of the three "
                         "block sets on three streams (profiling passes: one dispatch per step to count)")
    ap.add_argument("--preflight", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-preflight", action="store_true", help=argparse.SUPPRESS)  # tests: preflight with --same-device
    ap.add_argument("--force-rccl-leg", action="store_true", help=argparse.SUPPRESS)   # tests: the RCCL leg on gloo
    ap.add_argument("--preflight-limit-s", type=float, default=90.0, help=argparse.SUPPRESS)  # cap of the preflight child
    ap.add_argument("--rccl-leg-limit-s", type=float, default=90.0, help=argparse.SUPPRESS)   # cap of the RCCL leg
    ap.add_argument("--min-timed-ms", type=float, default=1000.0,
                    help="the timed call of --steps steps is repeated until the timed region lasts at least this long; "
                         "the legs of the N = 1 line time regions of the same length")
    ap.add_argument("--big-refine", type=int, default=38,
                    help="refinement n of the 8-GPU beam whose rank 3 of 8 / whole mesh the per_gpu_of_8 / cache_exceeding "
                         "legs of the N = 1 line step (38: BASELINE.json's 8.2M tets; smaller values are for tests)")
    ap.add_argument("--legs", default="all",
                    help="N = 1: comma-separated extra legs to run (roofline, per_gpu_of_8, cache_exceeding, unstructured, "
                         "predictor, cpu_baseline), or all / none")
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
    legs = Legs(job_start_time(), args.budget_s, rank)
    # N > 1: the direct peer exchange maps other processes' device memory - first use on this machine happens in a
    # child process, so that a fault there costs the child, not the benchmark (which then takes the RCCL all-reduce)
    exchange = "torch" if args.torch_exchange else args.exchange
    peer_ok = True
    if world > 1 and exchange == "auto" and (not args.same_device or args.force_preflight):
        limit = min(args.preflight_limit_s, 0.25 * max(legs.left(), 0.0))
        legs.begin("preflight")
        peer_ok = run_preflight(args, world, limit)
        legs.end("preflight", "done" if peer_ok else f"failed or exceeded {limit:.0f} s: all-reduce instead of the peer exchange")

    legs.begin("setup")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver
    from synchronization_avoiding_algorithms_amd.mesh import slab_partition

    if world > 1 and exchange == "auto":  # every rank takes the same transport
        flags = [None] * world
        dist.all_gather_object(flags, bool(peer_ok))
        if not all(flags):
            exchange = "rccl"
    preflight = None if world == 1 or args.exchange != "auto" or args.torch_exchange else bool(peer_ok)
    n = args.refine or N_FOR_GPUS.get(world, int(round((world * 1028850 / 150.0) ** (1 / 3))))
    mesh = bench_mesh(n, args.mesh)
    ne_total, nn_total = len(mesh.tets), len(mesh.points)
    epart = slab_partition(mesh, world) if world > 1 else np.zeros(ne_total, dtype=np.int64)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build_and_time(how, synced_graph=None):
        """Partition solver with the given transport, W warm-up and K timed steps; ``ok`` is False on any rank if a
        wait inside a kernel gave up (peer exchange: a neighbour's values never arrived)."""
        part = PartitionedSolver(mesh.points, mesh.tets, mesh.triangles, epart, rank, world, E=E, nu=NU, rho=RHO,
                                 fz=FZ, alpha=ALPHA, gamma=GAMMA, device=local_rank, block_nodes=args.block_nodes,
                                 threads=args.threads, exchange=how)
        if args.no_split_stepping and hasattr(part.solver, "set_option"):
            part.solver.set_option("split_stepping", 0)
        if synced_graph is not None and hasattr(part.solver, "set_option"):
            part.solver.set_option("synced_graph", synced_graph)  # saa_step_synced: replayed graphs / eager launches
        part.step_synced(args.warmup)  # world == 1: plain steps; else one exchange of shared-node forces per step
        # The timed call is `part.step_synced(args.steps)`.  Which kernels that runs depends on the call length (calls
        # of >= 8 steps take the resident kernel), so exactly that call is issued once more untimed - the first launch
        # of a kernel pays code-object upload - and its duration sizes the number of timed repetitions so that the
        # timed region lasts >= --min-timed-ms.
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

    legs.end("setup")
    legs.begin("headline")
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

    stats = sol.plan_stats()
    out = {
        "metric": "element_updates_per_s", "value": ne_total * args.steps / elapsed,
        "unit": "element-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "timed_calls": timed_calls,
        "ms_per_step": 1e3 * elapsed / args.steps, "steps_per_s": args.steps / elapsed,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("synthetic cantilever, Delaunay tetrahedra of random points in the 25 x 1 x 1 box at the "
                                f"node count of n={n}" if args.mesh == "delaunay" else
                                f"synthetic 25n x n x n Kuhn-tet cantilever n={n}"
                                f"{' (nodes jittered, numbering shuffled)' if args.mesh == 'jittered' else ''}") + ": "
                               f"{ne_total} tets, {nn_total} nodes, {world} x-slab partition(s), fp64, E=1e6 nu=0.3 "
                               f"alpha=0.5 ramped body force, dt={dt:.6e}",
                   "exchange": "none (1 partition)" if world == 1 else
                               {"peer": f"every step the fp64 forces of the shared nodes ({len(gshared)} in all) are "
                                        "stored into the neighbour ranks' memory (HIP IPC, xGMI peer stores) and "
                                        "summed in rank order (saa_step_peer); no collective",
                                "rccl": f"all-reduce of {3 * len(gshared)} fp64 shared-node forces every step, "
                                        "ncclAllReduce issued from C++ (saa_step_synced)",
                                "torch": f"all-reduce of {3 * len(gshared)} fp64 shared-node forces every step, "
                                         f"torch.distributed ({args.backend})"}[part.exchange],
                   "plan": stats, "kernel_sources": kernel_sources_digest()},
    }
    if world > 1:
        out["config"]["peer_preflight_rank0"] = preflight  # child-process trial of the peer exchange (None: not run)
        if retried:
            out["config"]["retried"] = retried
    legs.end("headline")
    with legs.lock:
        legs.line = out  # from here on the line is printed whatever happens (watchdog)

    def put(key, value):
        with legs.lock:
            out[key] = value

    # N > 1: the same partitions in sync-avoiding mode (BASELINE.json configs[4]; Online_predictor.py:251-318).
    rccl_wanted = (world > 1 and not args.no_rccl_leg and
                   ((part.exchange == "peer" and args.backend == "nccl" and not args.same_device) or
                    args.force_rccl_leg))
    rccl_reserve = 60.0 if rccl_wanted else 0.0
    if world > 1 and not args.no_sync_avoiding:
        # what the leg needs besides training: the recorded run, the predictor's capture, the windows (~20 s at N = 8)
        train_s = min(args.sa_train_seconds, legs.left() - rccl_reserve - 45.0)
        t = torch.tensor([train_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)  # one decision for all ranks
        train_s = float(t.item())
        if train_s < 5.0:
            legs.skip("sync_avoiding", f"{legs.left():.0f} s of the budget left")
        else:
            legs.begin("sync_avoiding")
            put("sync_avoiding", sync_avoiding_leg(part, args, rank, world, ne_total, fence, train_s))
            legs.end("sync_avoiding")

    if world == 1:
        single_gpu_legs(args, legs, put, sol, stats, n, ne_total, nn_total, local_rank)
    sol.close()  # (idempotent: the N = 1 legs have closed it already to make room for their own solvers)
    # N > 1: BASELINE.json configs[3] names the RCCL all-reduce as the per-step exchange.  When `value` above was
    # measured with the peer exchange, the same partitions are stepped once more with ncclAllReduce issued from C++
    # (saa_step_synced) and reported next to it.
    if rccl_wanted:
        limit = min(args.rccl_leg_limit_s, legs.left() - 12.0)
        t = torch.tensor([limit], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        limit = float(t.item())
        if limit < 20.0:
            legs.skip("rccl_allreduce", f"{legs.left():.0f} s of the budget left")
            put("rccl_allreduce", {"value": None, "exchange": "skipped: budget"})
        else:
            # a hung collective must not cost the headline line: past the limit the watchdog prints what has been
            # measured and leaves with a NON-ZERO status, so that the hang is seen
            put("rccl_allreduce", {"value": None, "exchange": f"timed out after {limit:.0f} s"})
            legs.begin("rccl_allreduce", limit_s=limit)
            test_hook("rccl_leg")
            k_r, w_r = min(args.steps, 2000), min(args.warmup, 200)
            args_steps, args_warmup = args.steps, args.warmup
            args.steps, args.warmup = k_r, w_r
            # twice: eager launches first (fused kernel, ncclAllReduce, finish kernel enqueued one by one - the route that
            # has run before), then the same steps as replayed HIP graphs of three steps each (saa_step_synced's default);
            # the first figure is in the line before the second route is tried
            res, ok_any = {}, False
            for route, env in (("eager", 0), ("graph", 1)):
                try:
                    part_r, elapsed_r, ok_r = build_and_time("rccl" if args.backend == "nccl" else "torch", synced_graph=env)
                    how = part_r.exchange
                    part_r.close()
                except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline line
                    ok_r, how, elapsed_r = False, f"failed: {e}"[:200], float("nan")
                res[route] = ({"value": ne_total * k_r / elapsed_r, "ms_per_step": 1e3 * elapsed_r / k_r, "exchange": how}
                              if ok_r else {"value": None, "exchange": how})
                ok_any = ok_any or ok_r
                best = max((r for r in res.values() if r["value"]), key=lambda r: r["value"], default=None)
                put("rccl_allreduce", dict(best or {"value": None, "exchange": how}, unit="element-updates/s", steps=k_r,
                                           warmup=w_r, routes=dict(res),
                                           note="same partitions; shared-node forces summed by an all-reduce of "
                                                f"{3 * len(gshared)} doubles every step instead of the peer exchange; "
                                                "value = the faster of the two launch routes"))
                if how != "rccl":  # (the torch transport has no graph route)
                    break
            args.steps, args.warmup = args_steps, args_warmup
            legs.end("rccl_allreduce", "done" if ok_any else "failed")
    legs.emit(final=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
