"""Two ranks sharing the one GPU of the test box: the real HIP begin / all-reduce / finish path
(gloo moves the CUDA interface buffer; the driver's multi-GPU bench uses nccl = RCCL) against the
reference's own 2-rank trajectory, and the predicted-phase overwrite / history kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO, free_port, load_golden, rel_l2

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, steps, out_dir, exchange):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver

    g = np.load(os.path.join(GOLDEN, "beam_coarse_mesh.npz"))
    t = np.load(os.path.join(GOLDEN, "tworank_trajectory.npz"))
    part = PartitionedSolver(g["points"], g["tetra"], g["triangle"], t["epart"], rank, world, device=0,
                             exchange=exchange)
    assert part.exchange == exchange, part.exchange  # no silent fallback in the test
    hist = torch.zeros((max(steps), part.input_size), dtype=torch.float64, device="cuda")
    snaps, done = {}, 0
    for s in steps:
        part.step_synced(s - done, hist, done)
        done = s
        snaps[s] = part.get_state()[0][:, 0]
    torch.cuda.synchronize()
    h = hist.cpu().numpy()
    # the recorded history is the shared dofs of every step (Online_predictor.py:260)
    assert np.array_equal(h[done - 1], snaps[done][part.layout.loc_dof_shared])
    # predicted phase: overwrite with a table row and record it (Online_predictor.py:298-301)
    table = torch.arange(3 * part.input_size, dtype=torch.float64, device="cuda").reshape(3, -1) * 1e-6
    hist2 = torch.zeros((3, part.input_size), dtype=torch.float64, device="cuda")
    part.step_predicted(3, table, 0, hist2, 0)
    d0 = part.get_state()[0][:, 0]
    assert np.array_equal(d0[part.layout.loc_dof_shared], table[2].cpu().numpy())
    assert torch.equal(hist2, table)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(k): v for k, v in snaps.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["peer", "torch"])
def test_two_ranks_on_one_gpu_match_reference(tmp_path, exchange):
    """``peer``: shared-node forces stored straight into the other rank's inbox through HIP IPC (the path the
    multi-GPU bench takes over xGMI); ``torch``: begin / all_reduce / finish."""
    steps = (1, 10, 100, 1000, 5000)
    port = free_port()
    mp.spawn(_worker, args=(2, port, steps, str(tmp_path), exchange), nprocs=2, join=True)
    t = load_golden("tworank_trajectory.npz")
    bound = {1: 1e-15, 10: 1e-14, 100: 1e-13, 1000: 5e-12, 5000: 5e-11}
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        for s in steps:
            err = rel_l2(got[str(s)], t[f"r{r}_step_{s}"])
            assert err < bound[s], (r, s, err)


def _workflow_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synchronization_avoiding_algorithms_amd import drivers
    from synchronization_avoiding_algorithms_amd.mesh import Mesh

    g = np.load(os.path.join(GOLDEN, "beam_coarse_mesh.npz"))
    h = np.load(os.path.join(GOLDEN, "hybrid_tworank.npz"))
    mesh = Mesh(g["points"], {"tetra": g["tetra"], "triangle": g["triangle"]})
    T, n_p, n_f, n_s, hid = (int(h[k]) for k in ("test_num", "n_past", "n_future", "filter_size", "hidden_size"))
    _, truth = drivers.data_prepare(mesh, T, 1, out_dir, rank, world, epart=h["epart"])
    dist.barrier()
    _, shared = drivers.shared_extraction(out_dir, rank)
    # the reference's weights at the path Online_predictor.py:139-140 builds
    mpath = os.path.join(out_dir, drivers.PATHS["model"].format(r=rank, nB=10, nH=hid, lr=5e-4, ns=n_s))
    os.makedirs(os.path.dirname(mpath), exist_ok=True)
    torch.save({k[len(f"r{rank}_w::"):]: torch.from_numpy(h[k]) for k in h.files if k.startswith(f"r{rank}_w::")},
               mpath)
    _, modeled, hist = drivers.online_predictor(mesh, T, 1, out_dir, rank, world, epart=h["epart"], n_past=n_p,
                                                n_future=n_f, filter_size=n_s, hidden_size=hid)
    np.savez(os.path.join(out_dir, f"wf{rank}.npz"), truth=truth, shared=shared, modeled=modeled,
             hist=hist.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_reference_workflow_on_gpu(tmp_path):
    """Data_prepare -> Shared_extraction -> Online_predictor (README.md:33-38) with the HIP solver and the
    GPU-batched LSTM, against the same chain run by the reference itself (hybrid_tworank.npz)."""
    port = free_port()
    mp.spawn(_workflow_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    h = load_golden("hybrid_tworank.npz")
    i_cri = int(h["n_past"]) * int(h["filter_size"]) - 1
    for r in range(2):
        got = np.load(tmp_path / f"wf{r}.npz")
        assert rel_l2(got["truth"][:, -1], h[f"r{r}_truth_last"]) < 1e-12
        assert rel_l2(got["shared"], h[f"r{r}_shared_traj"]) < 1e-12
        ref = h[f"r{r}_modeled"]
        assert rel_l2(got["modeled"][:, :i_cri + 1], ref[:, :i_cri + 1]) < 1e-12
        assert rel_l2(got["modeled"], ref) < 1e-4      # fp32 LSTM on MIOpen/rocBLAS vs CPU batch-1
        assert rel_l2(got["hist"], h[f"r{r}_d_sol_shared"]) < 1e-4
        # artefact names of the reference
        for key in ("local_nodes", "shared", "elements", "truth", "modeled", "shared_traj"):
            base = os.path.splitext(os.path.join(tmp_path, drivers_path(key, r)))[0]
            assert any(os.path.exists(base + ext) for ext in (".csv", ".hdf5", ".npz")), base


def _resync_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synchronization_avoiding_algorithms_amd import drivers
    from synchronization_avoiding_algorithms_amd import predictor as pr
    from synchronization_avoiding_algorithms_amd.mesh import Mesh

    g = np.load(os.path.join(GOLDEN, "beam_coarse_mesh.npz"))
    h = np.load(os.path.join(GOLDEN, "hybrid_tworank.npz"))
    mesh = Mesh(g["points"], {"tetra": g["tetra"], "triangle": g["triangle"]})
    T, n_p, n_f, n_s, hid = (int(h[k]) for k in ("test_num", "n_past", "n_future", "filter_size", "hidden_size"))
    model = pr.LSTM_encoder_decoder(int(h[f"r{rank}_loc_dof_shared"].size), hid)
    model.load_state_dict({k[len(f"r{rank}_w::"):]: torch.from_numpy(h[k]) for k in h.files if k.startswith(f"r{rank}_w::")})
    # (on the GPU the trajectory is recorded by the step kernels themselves: whole windows and whole re-synchronisations
    # per call, drivers._device_recorder)
    _, modeled, hist = drivers.online_predictor(mesh, T, 1, out_dir, rank, world, epart=h["epart"], n_past=n_p,
                                                n_future=n_f, filter_size=n_s, hidden_size=hid, model=model,
                                                scale=tuple(float(v) for v in h[f"r{rank}_scale"]),
                                                resync_every=2, resync_steps=7)
    out = dict(modeled=modeled, hist=hist.cpu().numpy())
    dist.barrier()
    np.savez(os.path.join(out_dir, f"rs{rank}.npz"), **out)
    dist.destroy_process_group()


def test_hybrid_loop_with_resynchronisation_on_gpu(tmp_path):
    """BASELINE.json configs[4]'s "RCCL every k-th step only" (an extension; the reference never synchronises again):
    Online_predictor with 7 synchronised steps after every 2 predicted windows, HIP solver + native predictor, two ranks,
    against the oracle's loop with the same schedule (tests/test_distributed_gloo.py has the CPU twin)."""
    from conftest import oracle_hybrid_tworank

    port = free_port()
    mp.spawn(_resync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    save, hist = oracle_hybrid_tworank(resync_every=2, resync_steps=7)
    h = load_golden("hybrid_tworank.npz")
    for r in range(2):
        got = np.load(tmp_path / f"rs{r}.npz")
        assert rel_l2(got["modeled"], save[r]) < 1e-4 and rel_l2(got["hist"], hist[r]) < 1e-4
        assert rel_l2(got["modeled"][:, :20], save[r][:, :20]) < 1e-12   # warm-up: fp64
        # the steps of a re-synchronisation start from predicted values and are fp64 from there: the history rows they
        # record are the shared dofs of the state they produced
        loc = h[f"r{r}_loc_dof_shared"]
        for i in (60, 66, 107, 113):
            assert np.array_equal(got["hist"][i], got["modeled"][loc, i])
        assert rel_l2(got["modeled"][:, :60], h[f"r{r}_modeled"][:, :60]) < 1e-4   # the reference's own run until step 60


def drivers_path(key, r):
    from synchronization_avoiding_algorithms_amd import drivers

    return drivers.PATHS[key].format(r=r)


def test_native_rccl_exchange_single_rank(beam_coarse):
    """saa_comm_init / saa_step_synced with a one-rank RCCL communicator: declared-shared nodes take the
    begin -> ncclAllReduce -> finish route and must land where plain steps do."""
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes

    mesh = beam_coarse
    lmd, mu = fs.lame(1e6, 0.3)
    layouts, _ = fs.build_layouts(mesh.tets, np.zeros(len(mesh.tets), dtype=int), 1, len(mesh.points),
                                  clamp_nodes(mesh))
    lay = layouts[0]
    lumped, fpre = fs.lumped_mass_and_load(mesh.points, mesh.tets, 1.0, 0.5)
    dt = fs.cfl_dt(mesh.points, mesh.tets, 1e6, 0.3, 1.0, 0.9)
    shared = np.array([3, 17, 40, 41, 77, 100, 5, 60], dtype=np.int32)
    args = (mesh.points[lay.nodes], lay.cells_local, lumped[lay.local_dof], fpre[lay.local_dof],
            lay.dirichlet_dofs, lmd, mu, dt, 0.5)
    plain = saa.HipExplicitSolver(*args)
    synced = saa.HipExplicitSolver(*args, shared_local=shared, shared_slots=np.arange(8, dtype=np.int32)[::-1].copy(),
                                   n_global_shared=10)  # two slots belong to "other ranks" and stay zero
    iface = torch.zeros(30, dtype=torch.float64, device="cuda")
    hist = torch.zeros((200, 24), dtype=torch.float64, device="cuda")
    synced.set_interface_buffer(iface)
    synced.set_stream(torch.cuda.current_stream().cuda_stream)
    synced.comm_init(synced.comm_unique_id(), 0, 1)
    plain.step(200)
    synced.step_synced(200, hist, 0)
    torch.cuda.synchronize()
    a, b = plain.get_state()[0], synced.get_state()[0]
    assert rel_l2(b, a) < 1e-13
    dof = (3 * shared[:, None] + np.arange(3)[None, :]).ravel()
    assert np.array_equal(hist[199].cpu().numpy(), b[dof, 0])
    assert float(iface[24:].abs().max()) == 0.0  # foreign slots re-zeroed
    # without history rows the same steps run as replayed HIP graphs (three steps per graph, the clock and with it the
    # ramp in device memory): 2 x 100 more steps - graphs of both rotation phases plus eager remainders - against plain
    # steps and against the eager route of a second handle
    eager = saa.HipExplicitSolver(*args, shared_local=shared, shared_slots=np.arange(8, dtype=np.int32)[::-1].copy(),
                                  n_global_shared=10)
    iface2 = torch.zeros(30, dtype=torch.float64, device="cuda")
    eager.set_interface_buffer(iface2)
    eager.set_stream(torch.cuda.current_stream().cuda_stream)
    eager.comm_init(eager.comm_unique_id(), 0, 1)
    eager.set_option("synced_graph", 0)
    eager.step_synced(400)
    plain.step(200)
    synced.step_synced(100)
    synced.step_synced(100)
    torch.cuda.synchronize()
    (a, an, ta), (b, bn, tb), (c, cn, tc) = plain.get_state(), synced.get_state(), eager.get_state()
    assert ta == tb == tc  # the host's clock advanced exactly like the device's
    assert rel_l2(b, a) < 1e-13 and rel_l2(bn, an) < 1e-13 and rel_l2(b, c) < 1e-13 and rel_l2(bn, cn) < 1e-13
    # The library's graphs after many other launches: PyTorch's captured validation graph returned corrupted sums once
    # ~10^4 kernels had been launched since its capture - these must not.
    z = torch.zeros(1000, device="cuda")
    for _ in range(20000):
        z.add_(1.0)
    plain.step(201)
    synced.step_synced(201)  # 67 replays of the graphs instantiated above
    torch.cuda.synchronize()
    (a, an, ta), (b, bn, tb) = plain.get_state(), synced.get_state()
    assert ta == tb and rel_l2(b, a) < 1e-13 and rel_l2(bn, an) < 1e-13 and float(z[0]) == 20000.0
    plain.close()
    synced.close()
    eager.close()


def _nccl_single_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver

    g = np.load(os.path.join(GOLDEN, "beam_coarse_mesh.npz"))
    epart = np.zeros(len(g["tetra"]), dtype=np.int64)
    part = PartitionedSolver(g["points"], g["tetra"], g["triangle"], epart, 0, 1, device=0)
    # the same hand-shake bench.py / PartitionedSolver perform at world > 1: torch's RCCL communicator and the
    # library's own one (created from the SAME librccl.so) side by side
    uid = part.solver.comm_unique_id()
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    flag = torch.tensor([1], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    part.solver.set_interface_buffer(part.iface)
    part.solver.comm_init(box[0], 0, 1)
    part.solver.step_synced(50)
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)  # torch's communicator still works afterwards
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "nccl1.npy"), part.get_state()[0])
    part.close()
    dist.destroy_process_group()


def test_native_exchange_next_to_torch_nccl(tmp_path):
    """torch.distributed(nccl) and saa_comm_init use the same librccl.so in one process without disturbing
    each other (one rank; the multi-rank collective itself needs the driver's multi-GPU node)."""
    port = free_port()
    mp.spawn(_nccl_single_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "nccl1.npy")
    ref = load_golden("serial_trajectory.npz")
    # 50 steps lie between the golden snapshots; compare with a fresh plain run instead
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import Mesh, clamp_nodes

    g = load_golden("beam_coarse_mesh.npz")
    mesh = Mesh(g["points"], {"tetra": g["tetra"], "triangle": g["triangle"]})
    lmd, mu = fs.lame(1e6, 0.3)
    lay = fs.build_layouts(mesh.tets, np.zeros(len(mesh.tets), dtype=int), 1, len(mesh.points), clamp_nodes(mesh))[0][0]
    lumped, fpre = fs.lumped_mass_and_load(mesh.points, mesh.tets, 1.0, 0.5)
    sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, lumped[lay.local_dof], fpre[lay.local_dof],
                                lay.dirichlet_dofs, lmd, mu, fs.cfl_dt(mesh.points, mesh.tets, 1e6, 0.3, 1.0, 0.9), 0.5)
    sol.step(50)
    assert rel_l2(got, sol.get_state()[0]) < 1e-13 and ref is not None
    sol.close()


@pytest.mark.parametrize("exchange", ["peer", "torch"])
def test_three_ranks_on_one_gpu_equal_serial(tmp_path, exchange):
    """Same as the gloo CPU test, with the real kernels (three processes share the test GPU); with ``peer`` the
    triple-owned nodes receive two pushes each and are summed in rank order."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_distributed_gloo import _check_three_ranks

    _check_three_ranks(tmp_path, use_gpu=True, exchange=exchange)


def test_three_ranks_resident_kernel_with_triple_owned_nodes(tmp_path):
    """The PEER variant of the resident kernel where some nodes have three holders (two pushes per node, three-term
    rank-ordered sums) and the holders are three processes."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_distributed_gloo import _check_three_ranks

    _check_three_ranks(tmp_path, use_gpu=True, exchange="peer", force_resident=True)


def _resident_peer_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver

    g = np.load(os.path.join(GOLDEN, "beam_coarse_mesh.npz"))
    t = np.load(os.path.join(GOLDEN, "tworank_trajectory.npz"))
    part = PartitionedSolver(g["points"], g["tetra"], g["triangle"], t["epart"], rank, world, device=0, exchange="peer",
                             wait_timeout_s=20, resident_on_shared_device=True)
    assert part.exchange == "peer" and part.solver.resident_kernel_info()["capable"]
    hist = torch.zeros((100, part.input_size), dtype=torch.float64, device="cuda")
    part.step_synced(1, hist, 0)     # one launch per step
    part.step_synced(9, hist, 1)     # resident kernel: pushes / collects between two processes inside the step loop
    d10 = part.get_state()[0][:, 0]
    part.step_synced(90, hist, 10)
    d100 = part.get_state()[0][:, 0]
    torch.cuda.synchronize()
    assert np.array_equal(hist[99].cpu().numpy(), d100[part.layout.loc_dof_shared])
    np.savez(os.path.join(out_dir, f"res{rank}.npz"), d10=d10, d100=d100)
    dist.barrier()
    dist.destroy_process_group()


def test_resident_kernel_with_peer_exchange_between_processes(tmp_path):
    """The PEER variant of the resident kernel with a real second process (both on the test GPU; the two small
    cooperative kernels only advance each other by time-slicing, so the step count is kept low)."""
    port = free_port()
    mp.spawn(_resident_peer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t = load_golden("tworank_trajectory.npz")
    for r in range(2):
        got = np.load(tmp_path / f"res{r}.npz")
        assert rel_l2(got["d10"], t[f"r{r}_step_10"]) < 1e-14
        assert rel_l2(got["d100"], t[f"r{r}_step_100"]) < 1e-13


def _dead_peer_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synchronization_avoiding_algorithms_amd.distributed import PartitionedSolver

    g = np.load(os.path.join(GOLDEN, "beam_coarse_mesh.npz"))
    t = np.load(os.path.join(GOLDEN, "tworank_trajectory.npz"))
    part = PartitionedSolver(g["points"], g["tetra"], g["triangle"], t["epart"], rank, world, device=0, exchange="peer",
                             wait_timeout_s=0.3)
    assert part.exchange == "peer"
    part.step_synced(3)
    part.get_state()
    dist.barrier()
    outcome = "idle"
    if rank == 0:  # rank 1 never takes this step: the wait inside the exchange kernel must give up, not hang
        part.step_synced(1)
        try:
            part.get_state()
            outcome = "no error"
        except RuntimeError as e:
            outcome = str(e)
    dist.barrier()
    with open(os.path.join(out_dir, f"dead{rank}.txt"), "w") as fh:
        fh.write(outcome)
    dist.destroy_process_group()


def test_peer_exchange_times_out_instead_of_hanging(tmp_path):
    port = free_port()
    mp.spawn(_dead_peer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert "timed out waiting for a neighbour" in (tmp_path / "dead0.txt").read_text()
