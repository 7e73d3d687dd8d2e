"""The reference's driver scripts as functions + CLIs (one process per GPU).

``Data_prepare.py`` -> :func:`data_prepare`, ``Shared_extraction.py`` -> :func:`shared_extraction`,
``Online_predictor.py`` -> :func:`online_predictor`; same artefact names under ``Results/`` and
``Distributed_save/``, same constants by default.  Launch like the reference's
``mpirun -np P python3 X.py``:

    python -m torch.distributed.run --nproc-per-node P --master-addr 127.0.0.1 \\
        -m synchronization_avoiding_algorithms_amd.drivers data_prepare --mesh Mesh_info/beam_coarse.vtk

Differences, all forced by the environment or by scale: the element partition comes from
``mesh.slab_partition`` / ``rcb_partition`` (ParMETIS is not available); the steady solve of
``Data_prepare.py:158-168`` (a dense O(N^3) diagnostic that never feeds the time loop) is not run; the
ghost step is the exact zero the reference obtains for a ramped load (``Data_prepare.py:178-191``).
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import predictor as pr
from . import results_io as rio
from .distributed import PartitionedSolver, run_hybrid
from .mesh import rcb_partition, read_vtk, slab_partition, structured_beam

# constants of Data_prepare.py:35-50 / Online_predictor.py:38-63
DEFAULTS = dict(E=1e6, nu=0.3, rho=1.0, fz=0.5, alpha=0.5, gamma=0.9)
STEADY_PATH = "Results/Static/steady_distributed.vtk"   # Data_prepare.py:25,168
PATHS = dict(local_nodes="Results/Rankwised_Data/Rank={r}_local_nodes.csv",
             shared="Results/Shared_Data/Rank={r}_shared.csv",
             global_shared="Results/Shared_Data/Global_shared.csv",
             elements="Results/Rankwised_Element/Rank={r}_elements.csv",
             truth="Results/Dynamics/Local-rank-{r}.hdf5",
             modeled="Results/Dynamics/Modeled_Local-rank-{r}.hdf5",
             shared_traj="Results/sol_on_shared/rank={r}-shared_dof.hdf5",
             model="Distributed_save/Rank-{r}/nB-{nB}-nH-{nH}-Lr-{lr}-filter={ns}/model.pth")


def _dist_env():
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        import torch

        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend)
    return rank, world, local


def make_partition(mesh, world, how="slab"):
    if world == 1:
        return np.zeros(len(mesh.tets), dtype=np.int64)
    return slab_partition(mesh, world) if how == "slab" else rcb_partition(mesh, world)


# Largest trajectory matrix kept on the GPU by the drivers (bytes); beyond it the host collects column by column.
DEVICE_TRAJECTORY_BUDGET = 32 << 30


def _device_recorder(part, n_steps, save_every):
    """``d1_save`` (``Data_prepare.py:219,236-240``) as a device matrix filled by the step kernels themselves
    (``saa_set_recorder``), or None when the solver has no recorder / the matrix would not fit the budget."""
    import torch

    n_cols = int(n_steps / save_every)
    n_dof = 3 * len(part.layout.nodes)
    if (part.tensor_device.type != "cuda" or not hasattr(part.solver, "set_recorder") or n_cols <= 0
            or 8 * n_dof * n_cols > DEVICE_TRAJECTORY_BUDGET):
        return None
    traj = torch.zeros((n_dof, n_cols), dtype=torch.float64, device=part.tensor_device)
    part.solver.set_recorder(traj, save_every, 0)
    return traj


def _saver(part, n_steps, save_every):
    store = np.zeros((3 * len(part.layout.nodes), int(n_steps / save_every)))
    state = {"counter": 0}

    def save(i, p):
        if i % save_every == 0 and state["counter"] < store.shape[1]:
            store[:, state["counter"]] = p.get_state()[0][:, 0]
            state["counter"] += 1

    return store, save


def data_prepare(mesh, n_steps=100000, save_every=1, out_dir=".", rank=0, world=1, partition="slab",
                 device=0, verbose=False, epart=None, **part_kw):
    """``Data_prepare.py:82-246``: partition, artefact CSVs, synchronised explicit run, trajectory file."""
    epart = make_partition(mesh, world, partition) if epart is None else epart
    part = PartitionedSolver(mesh.points, mesh.tets, mesh.triangles, epart, rank, world, device=device,
                             **{**DEFAULTS, **part_kw})
    lay = part.layout
    rio.save_int_list(os.path.join(out_dir, PATHS["shared"].format(r=rank)), lay.shared_nodes)
    rio.save_int_list(os.path.join(out_dir, PATHS["local_nodes"].format(r=rank)), lay.nodes)
    rio.save_int_list(os.path.join(out_dir, PATHS["elements"].format(r=rank)), lay.elements)
    if rank == 0:
        rio.save_int_list(os.path.join(out_dir, PATHS["global_shared"]), part.global_shared)
        if verbose:
            print("Time-step size is: " + str(part.dt))
    traj = _device_recorder(part, n_steps, save_every)
    if traj is not None:  # the kernels fill the trajectory; the whole run is a handful of launches
        part.step_synced(n_steps)
        part.solver.synchronize()
        store = traj.cpu().numpy()
        part.solver.set_recorder(None)
    else:
        store, save = _saver(part, n_steps, save_every)
        i = 0
        while i < n_steps:  # saved steps are i % save_every == 0 (Data_prepare.py:238-240)
            n = 1 if i % save_every == 0 else min(save_every - i % save_every, n_steps - i)
            part.step_synced(n)
            i += n
            save(i - 1, part)
    path = rio.save_displacement(os.path.join(out_dir, PATHS["truth"].format(r=rank)), store)
    part.close()
    return path, store


def steady_state(mesh, out_dir=".", device=0, tol=1e-12, verbose=False, E=None, nu=None, rho=None, fz=None):
    """``Data_prepare.py:157-168`` (rank 0 in the reference): the steady solution ``d = K^-1 F`` of the whole mesh
    under the un-ramped load, written as point data of ``Results/Static/steady_distributed.vtk``.  Matrix-free
    preconditioned CG on one GPU (:mod:`steady`) instead of the dense solve."""
    from . import fem_setup as fs
    from .mesh import clamp_nodes
    from .solver import HipExplicitSolver
    from .steady import steady_solve, stiffness_diagonal, write_vtk_point_data

    E = DEFAULTS["E"] if E is None else E
    nu = DEFAULTS["nu"] if nu is None else nu
    rho = DEFAULTS["rho"] if rho is None else rho
    fz = DEFAULTS["fz"] if fz is None else fz
    lmd, mu = fs.lame(E, nu)
    lumped, fpre, min_edge = fs.device_setup_fields(mesh.points, mesh.tets, rho, fz, device)
    dirichlet = fs.node_to_dof(clamp_nodes(mesh))
    sol = HipExplicitSolver(mesh.points, mesh.tets, lumped, fpre, dirichlet, lmd, mu,
                            fs.dt_from_min_edge(min_edge, E, nu, rho, DEFAULTS["gamma"]), DEFAULTS["alpha"],
                            device=device)
    d, iters, rel = steady_solve(sol, fpre, dirichlet, diag=stiffness_diagonal(mesh.points, mesh.tets, lmd, mu), tol=tol)
    sol.close()
    if verbose:
        print(f"steady solve: {iters} CG iterations, relative residual {rel:.2e}, max|d| = {np.abs(d).max():.6e}")
    path = write_vtk_point_data(os.path.join(out_dir, STEADY_PATH), mesh.points, mesh.tets, d)
    return path, d


def shared_extraction(out_dir=".", rank=0):
    """``Shared_extraction.py:22-40``: rows ``shared_dof`` of the rank's trajectory."""
    local = rio.load_int_list(os.path.join(out_dir, PATHS["local_nodes"].format(r=rank)))
    shared = rio.load_int_list(os.path.join(out_dir, PATHS["shared"].format(r=rank)))
    pos = {int(g): i for i, g in enumerate(local)}
    loc = np.array([pos[int(g)] for g in shared], dtype=np.int64)
    shared_dof = (3 * loc[:, None] + np.arange(3)[None, :]).ravel()
    data = rio.load_displacement(os.path.join(out_dir, PATHS["truth"].format(r=rank)))
    d = data[shared_dof, :]
    return rio.save_displacement(os.path.join(out_dir, PATHS["shared_traj"].format(r=rank)), d, compress=False), d


def online_predictor(mesh, n_steps=100000, save_every=1, out_dir=".", rank=0, world=1, partition="slab",
                     device=0, n_past=20, n_future=20, filter_size=150, hidden_size=50, nB=10,
                     learning_rate=5e-4, cut_off=0.5, model=None, scale=None, epart=None, resync_every=None,
                     resync_steps=None, **part_kw):
    """``Online_predictor.py:116-324``: warm-up with synchronisation, then LSTM-predicted halos.  ``resync_every`` /
    ``resync_steps``: synchronised steps again after every so many predicted windows (an extension, see
    :func:`distributed.run_hybrid`; None = the reference, which never synchronises again)."""
    import torch

    epart = make_partition(mesh, world, partition) if epart is None else epart
    part = PartitionedSolver(mesh.points, mesh.tets, mesh.triangles, epart, rank, world, device=device,
                             **{**DEFAULTS, **part_kw})
    if scale is None:  # Online_predictor.py:130-136
        traj = rio.load_displacement(os.path.join(out_dir, PATHS["shared_traj"].format(r=rank)))
        scale = pr.scaling_constants(traj, filter_size, n_past, n_future, cut_off)
    if model is None:  # Online_predictor.py:139-141
        mpath = os.path.join(out_dir, PATHS["model"].format(r=rank, nB=nB, nH=hidden_size, lr=learning_rate,
                                                            ns=filter_size))
        model = pr.call_model(part.tensor_device, filter_size, part.input_size, hidden_size, mpath)
    model = model.to(part.tensor_device)
    traj = _device_recorder(part, n_steps, save_every)
    store, save = (None, None) if traj is not None else _saver(part, n_steps, save_every)
    with torch.no_grad():
        hist = run_hybrid(part, n_steps, pr.DevicePredictor(model, n_past, n_future, filter_size, *scale),
                          n_past, n_future, filter_size, save=save, resync_every=resync_every,
                          resync_steps=resync_steps)
    if traj is not None:
        part.solver.synchronize()
        store = traj.cpu().numpy()
        part.solver.set_recorder(None)
    path = rio.save_displacement(os.path.join(out_dir, PATHS["modeled"].format(r=rank)), store)
    part.close()
    return path, store, hist


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def _load_mesh(args):
    return structured_beam(args.synthetic) if args.synthetic else read_vtk(args.mesh)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="synchronization_avoiding_algorithms_amd.drivers")
    ap.add_argument("command", choices=["data_prepare", "steady_state", "shared_extraction", "model_training",
                                        "online_predictor"])
    ap.add_argument("--epochs", type=int, default=None, help="model_training: override the epoch count")
    ap.add_argument("--mesh", default="Mesh_info/beam_coarse.vtk")
    ap.add_argument("--synthetic", type=int, default=0, help="use the 25n x n x n synthetic beam instead")
    ap.add_argument("--steps", type=int, default=100000)      # test_num, Data_prepare.py:49
    ap.add_argument("--save-every", type=int, default=1)      # Data_prepare.py:50
    ap.add_argument("--out", default=".")
    ap.add_argument("--partition", choices=["slab", "rcb"], default="slab")
    ap.add_argument("--n-past", type=int, default=20)
    ap.add_argument("--n-future", type=int, default=20)
    ap.add_argument("--filter-size", type=int, default=150)
    ap.add_argument("--hidden-size", type=int, default=50)
    ap.add_argument("--resync-every", type=int, default=None,
                    help="online_predictor: synchronised steps again after every so many predicted windows (extension; "
                         "default: never, like the reference)")
    ap.add_argument("--resync-steps", type=int, default=None, help="how many (default: one window, n_future*filter_size)")
    args = ap.parse_args(argv)
    rank, world, local = _dist_env()
    if args.command == "data_prepare":
        path, _ = data_prepare(_load_mesh(args), args.steps, args.save_every, args.out, rank, world,
                               args.partition, device=local, verbose=True)
    elif args.command == "steady_state":
        if rank != 0:
            return
        path, _ = steady_state(_load_mesh(args), args.out, device=local, verbose=True)
    elif args.command == "shared_extraction":
        path, _ = shared_extraction(args.out, rank)
    elif args.command == "model_training":
        from .training import train_rank_model

        path, _, _ = train_rank_model(args.out, rank, device=f"cuda:{local}" if _has_gpu() else "cpu",
                                      hidden_size=args.hidden_size, filter_size=args.filter_size,
                                      n_past=args.n_past, n_future=args.n_future, num_epochs=args.epochs,
                                      verbose=True)
    else:
        path, _, _ = online_predictor(_load_mesh(args), args.steps, args.save_every, args.out, rank, world,
                                      args.partition, device=local, n_past=args.n_past, n_future=args.n_future,
                                      filter_size=args.filter_size, hidden_size=args.hidden_size,
                                      resync_every=args.resync_every, resync_steps=args.resync_steps)
    print(f"[rank {rank}] wrote {path}")


if __name__ == "__main__":
    main()
