"""ORACLE (test infrastructure, NOT product code) - the reference's distributed CPU path on P host cores.

What ``mpirun -np P python3 Data_prepare.py`` does per time step (/root/reference): every rank multiplies its assembled
CSR stiffness with its displacement (``Tools/Dynamic_solver.py:12``), the forces of nodes held by several ranks are
summed (``syn_cpus``, ``Tools/Distributed_tools.py:77-92``) and the NumPy update expression is evaluated
(``Dynamic_solver.py:29-32``).  Here: one process per rank (mpi4py is absent from this image; the ranks meet in a
shared-memory segment, the transport an MPI library would use on one node), the same SciPy / NumPy operations per step
(``fem_oracle.RankProblem`` / ``cd_update``), and the sum restricted to what it changes - the shared dofs, packed in the
sorted ``Global_shared`` order, added in rank order - instead of the reference's gather + broadcast of the whole global
vector (which would only make the baseline slower).  The rank processes import neither torch nor HIP.

Used by ``bench.py``'s ``cpu_baseline`` leg (P = min(8, host cores)) and by ``tests/test_distributed_gloo.py``, which
checks it against ``fem_oracle.explicit_step_synced``.  Never imported by the product package.
"""
from __future__ import annotations

import time

import numpy as np


def _rank_main(rank, world, token, mesh_n, steps, out_path, want_state):
    """One rank.  No torch, no HIP: these processes must not open the GPU (the box limits how many may)."""
    from multiprocessing import resource_tracker, shared_memory

    from oracle import fem_oracle as fo
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, slab_partition, structured_beam

    mesh = structured_beam(mesh_n)
    epart = slab_partition(mesh, world)
    lmd, mu = fo.lame(1e6, 0.3)
    # Data_prepare.py:127-136, 175-176, 147: clamp, lumped mass / load, CFL step (the vectorised NumPy closed forms give
    # the same numbers as the oracle's element loops - tests/test_oracle_golden.py - and keep the set-up short)
    lumped, fpre = fs.lumped_mass_and_load(mesh.points, mesh.tets, 1.0, 0.5)
    dt = fs.cfl_dt(mesh.points, mesh.tets, 1e6, 0.3, 1.0, 0.9)
    rp = fo.RankProblem(rank, epart, mesh.tets, mesh.points, clamp_nodes(mesh), lumped, fpre, lmd, mu)
    # every rank works the (replicated) mesh's bookkeeping out for itself, like Data_prepare.py:104-124 after its bcasts
    count = np.zeros(len(mesh.points), dtype=np.int32)
    for q in range(world):
        count[np.unique(mesh.tets[epart == q])] += 1
    gshared = np.nonzero(count > 1)[0]                                    # sort_shared (Distributed_tools.py:44-51)
    slot = np.full(len(mesh.points), -1, dtype=np.int64)
    slot[gshared] = np.arange(len(gshared))
    mine = np.nonzero(slot[rp.nodes] >= 0)[0]                             # local ids of my shared nodes
    loc_dof = fo.node_to_dof(mine)
    buf_dof = fo.node_to_dof(slot[rp.nodes[mine]])
    # The per-step sum of the shared-node forces: rank r writes its partial forces into row r of a (P, 3|Global_shared|)
    # array in shared memory, everybody waits until all rows of this step are there, and adds the rows IN RANK ORDER -
    # the summation order of syn_cpus (Distributed_tools.py:84-86).  Shared memory + a spin barrier stand in for
    # mpi4py's shared-memory transport; two buffers alternate so that one barrier per step suffices.
    width = max(3 * len(gshared), 1)
    n_bytes = 8 * (2 * world * width + world)
    name = f"saa_cpu_baseline_{token}"
    if rank == 0:
        shm = shared_memory.SharedMemory(name=name, create=True, size=n_bytes)
        np.ndarray((n_bytes // 8,), dtype=np.float64, buffer=shm.buf)[:] = 0.0
        np.ndarray((world,), dtype=np.int64, buffer=shm.buf, offset=8 * 2 * world * width)[0] = -1  # "segment is ready"
    else:
        deadline = time.time() + 300
        while True:
            try:
                shm = shared_memory.SharedMemory(name=name)
                if shm.size >= n_bytes and np.ndarray((world,), dtype=np.int64, buffer=shm.buf,
                                                      offset=8 * 2 * world * width)[0] == -1:
                    break
                shm.close()
            except FileNotFoundError:
                pass
            if time.time() > deadline:
                raise RuntimeError("cpu baseline: rank 0 never created the shared segment")
            time.sleep(0.01)
        resource_tracker.unregister(shm._name, "shared_memory")  # rank 0 owns the segment and unlinks it
    rows = np.ndarray((2, world, width), dtype=np.float64, buffer=shm.buf)
    arrive = np.ndarray((world,), dtype=np.int64, buffer=shm.buf, offset=8 * 2 * world * width)
    d0 = np.zeros((len(rp.local_dof), 1))
    dn = np.zeros_like(d0)
    tn = 0
    # start line: ranks > 0 report -2 in their slot, rank 0 waits for all of them and then sets every slot to 0
    if rank == 0:
        while not all(arrive[r] == -2 for r in range(1, world)):
            pass
        arrive[:] = 0
    else:
        arrive[rank] = -2
        while arrive[rank] != 0:
            pass
    t0 = time.perf_counter()
    for s in range(1, steps + 1):
        f = rp.K.dot(d0)                                                  # Dynamic_solver.py:12
        half = rows[s & 1]
        half[rank, buf_dof] = f[loc_dof, 0]
        arrive[rank] = s
        while arrive.min() < s:                                           # spin barrier (one core per rank)
            pass
        total = np.zeros(width)
        for r in range(world):                                            # rank order, like the root's loop
            total[buf_dof] += half[r, buf_dof]
        f[loc_dof, 0] = total[buf_dof]
        d1 = fo.cd_update(f, rp.F, rp.l_M, d0, dn, dt, tn, 0.5, rp.dirichlet)  # Dynamic_solver.py:29-32
        dn, d0 = d0, d1
        tn = tn + dt
    elapsed = time.perf_counter() - t0
    if want_state:
        np.save(f"{out_path}.rank{rank}.npy", d0)
    arrive[rank] = steps + 1                                              # done with the segment
    if rank == 0:
        while arrive.min() < steps + 1:
            pass
        np.save(out_path, np.array([elapsed, len(mesh.tets), len(gshared)], dtype=np.float64))
    del rows, arrive, half
    shm.close()
    if rank == 0:
        shm.unlink()


def run(world, mesh_n, steps, want_state=False, timeout=600):
    """Steps the ``25n x n x n`` beam in ``world`` x-slabs on ``world`` processes (fresh interpreters that never touch
    a GPU); returns ``{"seconds", "n_tets", "n_shared", "states"}``."""
    import os
    import subprocess
    import sys
    import tempfile

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    token = f"{os.getpid()}_{int(time.time() * 1e6) % 10 ** 12}"
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "result.npy")
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline_mp", str(r), str(world), token,
                                   str(mesh_n), str(steps), out, str(int(want_state))], cwd=repo, env=env)
                 for r in range(world)]
        deadline = time.time() + timeout
        try:
            for p in procs:
                if p.wait(timeout=max(1.0, deadline - time.time())) != 0:
                    raise RuntimeError(f"cpu baseline: a rank process exited with status {p.returncode}")
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        seconds, n_tets, n_shared = np.load(out)
        states = [np.load(f"{out}.rank{r}.npy") for r in range(world)] if want_state else None
    return {"seconds": float(seconds), "n_tets": int(n_tets), "n_shared": int(n_shared), "states": states}


if __name__ == "__main__":
    import sys

    a = sys.argv[1:]
    _rank_main(int(a[0]), int(a[1]), a[2], int(a[3]), int(a[4]), a[5], bool(int(a[6])))
