"""GPU parity at BASELINE.json's full sizes, through the very plans bench.py runs.

* config 2 - the synthetic ~1M-tet cantilever (n = 19: 1 028 850 tets, 190 400 nodes) on one GPU: 256 blocks x 1024
  threads, the resident multi-step kernel with its ~128 KB LDS image and multi-round item lists, and the fused
  one-launch-per-step kernel;
* configs 3/4 - the per-GPU workload of the 8-GPU run: the middle x-slab (rank 3 of 8) of the ~8M-tet cantilever
  (n = 38: 1 031 016 tets, 182 520 nodes, 3 042 shared nodes on two interface planes), stepped through the direct peer
  exchange (push / stamped entries / poll / rank-ordered sums, fused and resident PEER kernels) with the loop-back
  attach: its one imaginary neighbour returns this rank's own partial force, i.e. shared nodes are updated with
  exactly 2 x their local K_r d (saa_hip.h: saa_peer_attach_loopback), which the oracle reproduces.

No reference run exists at these sizes (its set-up is dense O(N^2..N^3)): the oracle here is
``fem_oracle.MatrixFreeStiffness`` - the reference's element matrices applied element by element - pinned to the
reference's own ``LocalK.dot`` on beam_coarse in tests/test_oracle_golden.py; plus size-independent properties (rigid
translation and infinitesimal rotation produce no force).

Tolerances (fp64): K.d rel-L2 < 1e-13; 200 steps (100 on the N = 2 / N = 4 slabs) from a rough state rel-L2 < 1e-11 (both
kernels, against the oracle and against each other); rigid modes: max|f| < 1e-12 x max|K.d_rand| for displacements of the
same size.  Every stepping test of this file compares with the oracle - since round 4 also the 8.2M-tet beam on one GPU (30 steps
in the suite, 100 outside it), next to its size-independent properties.
"""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

E, NU, RHO, FZ, ALPHA, GAMMA = 1e6, 0.3, 1.0, 0.5, 0.5, 0.9
N_STEPS = 200


def _build(mesh, n_parts, rank):
    """The partition exactly as bench.py builds it (same host code, automatic plan)."""
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, slab_partition

    lmd, mu = fs.lame(E, NU)
    epart = slab_partition(mesh, n_parts) if n_parts > 1 else np.zeros(len(mesh.tets), dtype=np.int64)
    lay, gshared, l_M, F, dt_local = fs.rank_problem(mesh.points, mesh.tets, clamp_nodes(mesh), epart, rank, n_parts, E,
                                                     NU, RHO, FZ, GAMMA, device=0)
    # dt = min over the ranks (Data_prepare.py:147-154) = the whole mesh's CFL step; here from the device kernel on all
    # elements, which must give the host Meshsize path's value bit for bit
    dt = fs.dt_from_min_edge(fs.device_setup_fields(mesh.points, mesh.tets, RHO, FZ)[2], E, NU, RHO, GAMMA)
    assert dt == fs.cfl_dt(mesh.points, mesh.tets, E, NU, RHO, GAMMA) and dt <= dt_local
    sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, l_M, F, lay.dirichlet_dofs, lmd, mu, dt, ALPHA,
                                shared_local=lay.shared_local, shared_slots=lay.shared_slots, n_global_shared=len(gshared))
    return sol, lay, dt, l_M, F, (lmd, mu)


@pytest.fixture(scope="module")
def beam38():
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(38)
    assert len(mesh.tets) == 8230800
    return mesh


@pytest.fixture(scope="module")
def middle_slab(beam38):
    """Rank 3 of the 8 x-slabs of the 8.2M-tet beam - the per-GPU workload of configs 3 / 4 - with the oracle's operator on
    it, built once for the tests that step it."""
    from oracle import fem_oracle as fo

    sol, lay, dt, l_M, F, (lmd, mu) = _build(beam38, 8, 3)
    sol.close()
    pts = beam38.points[lay.nodes]
    assert len(lay.cells_local) == 1031016 and len(lay.shared_local) == 3042
    K = fo.MatrixFreeStiffness(lay.cells_local, pts, lmd, mu)
    return {"lay": lay, "dt": dt, "l_M": l_M, "F": F, "lmd": lmd, "mu": mu, "pts": pts, "K": K}


def _slab_solver(ms):
    import synchronization_avoiding_algorithms_amd as saa

    lay = ms["lay"]
    return saa.HipExplicitSolver(ms["pts"], lay.cells_local, ms["l_M"], ms["F"], lay.dirichlet_dofs, ms["lmd"], ms["mu"],
                                 ms["dt"], ALPHA, shared_local=lay.shared_local, shared_slots=lay.shared_slots,
                                 n_global_shared=7 * 1521)


def _operator_properties(sol, K, pts, rng):
    d = rng.uniform(-1e-2, 1e-2, size=(sol.n_dof, 1))
    f_ref = K.dot(d)
    assert rel_l2(sol.internal_force(d), f_ref) < 1e-13
    scale = np.abs(f_ref).max()
    # rigid translation and infinitesimal rotation u = w x (x - c): zero strain, zero force (displacements <= 1e-2)
    t = np.tile([1.0e-2, -0.5e-2, 0.25e-2], sol.n_nodes)
    assert np.abs(sol.internal_force(t)).max() < 1e-12 * scale
    u = np.cross(np.array([3.0, -4.0, 5.0]), pts - pts.mean(axis=0))
    u *= 1e-2 / np.abs(u).max()
    assert np.abs(sol.internal_force(u.ravel())).max() < 1e-12 * scale
    # K is linear and symmetric (size-independent properties of the assembled operator, Mat_construction.py:79-150):
    # K(a d1 + b d2) = a K d1 + b K d2 and d2.K d1 = d1.K d2
    d2 = rng.uniform(-1e-2, 1e-2, size=(sol.n_dof, 1))
    f1, f2 = sol.internal_force(d), sol.internal_force(d2)
    assert rel_l2(sol.internal_force(0.75 * d - 2.5 * d2), 0.75 * f1 - 2.5 * f2) < 1e-13
    a, b = float(d2.ravel() @ f1.ravel()), float(d.ravel() @ f2.ravel())
    assert abs(a - b) < 1e-12 * max(abs(a), abs(b))
    assert np.abs(K.dot(u.ravel())).max() < 1e-12 * scale  # the oracle agrees that it is a null vector


def _rough_state(n_dof, dirichlet, rng):
    """Every dof moves from the first step on (a smooth start would leave most of the 200 steps near zero)."""
    d0 = rng.uniform(-1e-4, 1e-4, size=(n_dof, 1))
    dn = d0 + rng.uniform(-1e-6, 1e-6, size=(n_dof, 1))
    d0[dirichlet] = 0
    dn[dirichlet] = 0
    return d0, dn


def test_config2_one_million_tets_resident_and_fused_kernels():
    from oracle import fem_oracle as fo
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(19)
    assert len(mesh.tets) == 1028850 and len(mesh.points) == 190400
    sol, lay, dt, l_M, F, (lmd, mu) = _build(mesh, 1, 0)
    st, info = sol.plan_stats(), sol.resident_kernel_info()
    # the plan of the headline bench line: one 1024-thread workgroup per CU, resident kernel available
    assert st["n_blocks"] == 256 and st["threads"] == 1024 and info["capable"] and info["lds_bytes"] <= 160 * 1024
    K = fo.MatrixFreeStiffness(lay.cells_local, mesh.points[lay.nodes], lmd, mu)
    rng = np.random.default_rng(19)
    _operator_properties(sol, K, mesh.points[lay.nodes], rng)

    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    tn, o0, on = 0.25, d0, dn
    for _ in range(N_STEPS):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        on, o0, tn = o0, o1, tn + dt
    got = {}
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        sol.step(N_STEPS)
        g0, gn, gt = sol.get_state()
        assert gt == tn
        assert rel_l2(g0, o0) < 1e-11 and rel_l2(gn, on) < 1e-11, name
        got[name] = g0
    assert rel_l2(got["resident"], got["fused"]) < 1e-11
    assert np.abs(o0).max() > 1e-6
    sol.close()


def test_one_million_tets_with_jittered_nodes_and_shuffled_numbering():
    """The workload of ``bench.py --mesh jittered``: the same beam with every interior node moved by up to 20 % of a cell
    and nodes and elements numbered at random - the plan pairs its elements in a spatial order and gives the blocks a
    pseudo-lattice numbering with re-ordered halo lists (saa_plan.cpp), paths a lattice never takes.  Operator
    properties and 100 steps of both kernels against the oracle on that mesh (positive and distorted elements alike)."""
    import sys

    from conftest import REPO
    from oracle import fem_oracle as fo

    sys.path.insert(0, REPO)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        from bench import bench_mesh
    finally:
        sys.argv = argv
    mesh = bench_mesh(19, "jittered")
    sol, lay, dt, l_M, F, (lmd, mu) = _build(mesh, 1, 0)
    st = sol.plan_stats()
    assert st["n_blocks"] == 256 and sol.resident_kernel_info()["capable"]
    assert st["n_by_construction"] > 0.5 * st["n_items"] and st["n_pairs"] > 0.985 * st["n_elem_copies"] / 2, st
    K = fo.MatrixFreeStiffness(lay.cells_local, mesh.points[lay.nodes], lmd, mu)
    rng = np.random.default_rng(7)
    _operator_properties(sol, K, mesh.points[lay.nodes], rng)
    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    tn, o0, on = 0.25, d0, dn
    for _ in range(100):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        on, o0, tn = o0, o1, tn + dt
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        sol.step(100)
        g0, gn, gt = sol.get_state()
        assert gt == tn and rel_l2(g0, o0) < 1e-11 and rel_l2(gn, on) < 1e-11, name
    sol.close()


def test_one_million_tets_of_a_delaunay_mesh():
    """The workload of ``bench.py --mesh delaunay`` / the `unstructured` leg of the N = 1 line: the beam's box meshed by a
    Delaunay triangulation of random points (mesh.delaunay_beam: no lattice anywhere, valences 2...47, the class of mesh the
    reference's Gmsh input is, Mesh_info/beam_US.geo:2-16) - elements paired by augmenting paths, block nodes numbered
    while the LDS groups are formed (saa_plan.cpp: joint_pack_list).  Operator properties and 100 steps of both kernels
    against the oracle's element-by-element operator on that mesh."""
    from oracle import fem_oracle as fo
    from synchronization_avoiding_algorithms_amd.mesh import delaunay_beam

    mesh = delaunay_beam(19)
    assert len(mesh.points) == 190400 and len(mesh.tets) > 1000000
    sol, lay, dt, l_M, F, (lmd, mu) = _build(mesh, 1, 0)
    st = sol.plan_stats()
    assert st["n_blocks"] == 256 and sol.resident_kernel_info()["capable"]
    assert st["n_by_construction"] == 0 and st["n_renumbered"] == 256            # nothing a lattice numbering could use
    assert st["n_pairs"] > 0.97 * st["n_elem_copies"] / 2 and st["lds_conflict_factor"] < 1.5, st
    K = fo.MatrixFreeStiffness(lay.cells_local, mesh.points[lay.nodes], lmd, mu)
    rng = np.random.default_rng(11)
    _operator_properties(sol, K, mesh.points[lay.nodes], rng)
    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    d0, dn = 1e-2 * d0, 1e-2 * dn  # (elements down to 1/500 of the mean volume: keep the start inside the linear range)
    tn, o0, on = 0.25, d0, dn
    for _ in range(100):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        on, o0, tn = o0, o1, tn + dt
    assert np.isfinite(o0).all() and np.abs(o0).max() < 1.0
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        sol.step(100)
        g0, gn, gt = sol.get_state()
        assert gt == tn and rel_l2(g0, o0) < 1e-11 and rel_l2(gn, on) < 1e-11, name
    sol.close()


def test_config3_middle_slab_of_the_8gpu_partition_through_the_peer_exchange(middle_slab):
    import torch
    from oracle import fem_oracle as fo

    ms = middle_slab
    lay, dt, l_M, F, pts, K = ms["lay"], ms["dt"], ms["l_M"], ms["F"], ms["pts"], ms["K"]
    sol = _slab_solver(ms)
    n_sh = len(lay.shared_local)
    assert sol.resident_kernel_info()["capable"]
    rng = np.random.default_rng(38)
    _operator_properties(sol, K, pts, rng)  # the partial K_r of this rank, interface rows included

    sol.peer_attach_loopback(2)
    sh_dof = lay.loc_dof_shared
    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    tn, o0, on = 0.25, d0, dn
    hist_ref = np.zeros((N_STEPS, 3 * n_sh))
    for i in range(N_STEPS):
        f = K.dot(o0)
        f[sh_dof] = f[sh_dof] + f[sh_dof]  # rank 0's own force + the imaginary rank 1's copy of it, in rank order
        o1 = fo.cd_update(f, F, l_M, o0, on, dt, tn, ALPHA, lay.dirichlet_dofs)  # Dynamic_solver.py:26-32
        hist_ref[i] = o1[sh_dof, 0]                                              # Online_predictor.py:260
        on, o0, tn = o0, o1, tn + dt
    got = {}
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        hist = torch.zeros((N_STEPS, 3 * n_sh), dtype=torch.float64, device="cuda")
        sol.step_peer(N_STEPS, hist, 0)
        g0, gn, gt = sol.get_state()
        assert gt == tn
        assert rel_l2(g0, o0) < 1e-11 and rel_l2(gn, on) < 1e-11, name
        h = hist.cpu().numpy()
        assert np.array_equal(h[-1], g0[sh_dof, 0])
        assert rel_l2(h, hist_ref) < 1e-11, name
        got[name] = g0
    assert rel_l2(got["resident"], got["fused"]) < 1e-11
    sol.close()


def test_momentum_balance_of_the_free_one_million_tet_beam():
    """A size-independent property of the update (Tools/Dynamic_solver.py:13-32) that needs no oracle: without Dirichlet
    nodes the internal forces sum to zero (translation invariance of K), so for every step and component
        sum_nodes [ m (d1 - 2 d0 + dn) / dt^2 + alpha m (d1 - dn) / (2 dt) ] = ramp(tn) * sum_nodes F
    - checked on the columns the trajectory recorder takes out of one launch of the resident kernel."""
    import torch
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(19)
    lmd, mu = fs.lame(E, NU)
    lay, _ = fs.build_rank_layout(mesh.tets, np.zeros(len(mesh.tets), dtype=np.int64), 0, 1, len(mesh.points),
                                  np.zeros(0, dtype=np.int64))
    l_M, F, _ = fs.rank_fields(mesh.points, mesh.tets, lay, RHO, FZ)
    dt = fs.cfl_dt(mesh.points, mesh.tets, E, NU, RHO, GAMMA)
    sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, l_M, F, np.zeros(0, dtype=np.int32), lmd, mu, dt, ALPHA)
    assert sol.resident_kernel_info()["capable"]
    rng = np.random.default_rng(3)
    d0 = rng.uniform(-1e-4, 1e-4, size=(sol.n_dof, 1))
    dn = d0 + rng.uniform(-1e-6, 1e-6, size=(sol.n_dof, 1))
    t0 = 0.4
    sol.set_state(d0, dn, t0)
    n_cols = 12
    traj = torch.zeros((sol.n_dof, n_cols), dtype=torch.float64, device="cuda")
    sol.set_recorder(traj, save_every=1, next_step_index=0)
    sol.step(n_cols)  # one launch of the resident kernel
    torch.cuda.synchronize()
    cols = [dn.ravel(), d0.ravel()] + [traj[:, j].cpu().numpy() for j in range(n_cols)]
    m, f = np.asarray(l_M).ravel(), np.asarray(F).ravel()
    tn = t0
    for j in range(n_cols):
        a, b, c = cols[j], cols[j + 1], cols[j + 2]  # d^(n-1), d^n, d^(n+1)
        lhs = m * (c - 2 * b + a) / dt ** 2 + ALPHA * m * (c - a) / (2 * dt)
        for comp in range(3):
            want = min(tn, 1.0) * f[comp::3].sum()
            scale = np.abs(m[comp::3] * (c - 2 * b + a)[comp::3] / dt ** 2).sum()
            assert abs(lhs[comp::3].sum() - want) < 1e-9 * scale, (j, comp)
        tn = tn + dt
    sol.close()


def test_eight_million_tets_on_one_gpu_properties_of_the_fused_plan(beam38):
    """The cache-exceeding point of the bench (`--refine 38`: 8 230 800 tets on one GPU, 2048 blocks of 512 threads, the
    one-launch-per-step kernel) has no oracle run of its size; the same oracle-free properties as above hold for it:
    rigid modes give no force, K is linear and symmetric, and every step of the free beam balances momentum."""
    import torch
    import synchronization_avoiding_algorithms_amd as saa
    from synchronization_avoiding_algorithms_amd import fem_setup as fs

    mesh = beam38
    lmd, mu = fs.lame(E, NU)
    lay, _ = fs.build_rank_layout(mesh.tets, np.zeros(len(mesh.tets), dtype=np.int64), 0, 1, len(mesh.points),
                                  np.zeros(0, dtype=np.int64))
    l_M, F, _ = fs.rank_fields(mesh.points, mesh.tets, lay, RHO, FZ)
    dt = fs.cfl_dt(mesh.points, mesh.tets, E, NU, RHO, GAMMA)
    sol = saa.HipExplicitSolver(mesh.points[lay.nodes], lay.cells_local, l_M, F, np.zeros(0, dtype=np.int32), lmd, mu, dt, ALPHA)
    st = sol.plan_stats()
    assert st["n_blocks"] == 2048 and st["threads"] == 512 and not sol.resident_kernel_info()["capable"]
    pts = mesh.points[lay.nodes]
    rng = np.random.default_rng(38)
    d = rng.uniform(-1e-2, 1e-2, size=(sol.n_dof, 1))
    d2 = rng.uniform(-1e-2, 1e-2, size=(sol.n_dof, 1))
    f1, f2 = sol.internal_force(d), sol.internal_force(d2)
    scale = np.abs(f1).max()
    assert np.abs(sol.internal_force(np.tile([1.0e-2, -0.5e-2, 0.25e-2], sol.n_nodes))).max() < 1e-12 * scale
    u = np.cross(np.array([3.0, -4.0, 5.0]), pts - pts.mean(axis=0))
    u *= 1e-2 / np.abs(u).max()
    assert np.abs(sol.internal_force(u.ravel())).max() < 1e-12 * scale
    assert rel_l2(sol.internal_force(0.75 * d - 2.5 * d2), 0.75 * f1 - 2.5 * f2) < 1e-13
    a, b = float(d2.ravel() @ f1.ravel()), float(d.ravel() @ f2.ravel())
    assert abs(a - b) < 1e-12 * max(abs(a), abs(b))
    assert abs(f1.sum()) < 1e-9 * np.abs(f1).sum()  # internal forces of a free body sum to zero

    d0 = rng.uniform(-1e-4, 1e-4, size=(sol.n_dof, 1))
    dn = d0 + rng.uniform(-1e-6, 1e-6, size=(sol.n_dof, 1))
    t0 = 0.4
    sol.set_state(d0, dn, t0)
    n_cols = 10
    traj = torch.zeros((sol.n_dof, n_cols), dtype=torch.float64, device="cuda")
    sol.set_recorder(traj, save_every=1, next_step_index=0)
    sol.step(n_cols)
    torch.cuda.synchronize()
    cols = [dn.ravel(), d0.ravel()] + [traj[:, j].cpu().numpy() for j in range(n_cols)]
    m, f = np.asarray(l_M).ravel(), np.asarray(F).ravel()
    tn = t0
    for j in range(n_cols):
        p, q, r = cols[j], cols[j + 1], cols[j + 2]
        lhs = m * (r - 2 * q + p) / dt ** 2 + ALPHA * m * (r - p) / (2 * dt)
        for comp in range(3):
            want = min(tn, 1.0) * f[comp::3].sum()
            assert abs(lhs[comp::3].sum() - want) < 1e-9 * np.abs(m[comp::3] * (r - 2 * q + p)[comp::3] / dt ** 2).sum(), (j, comp)
        tn = tn + dt
    sol.close()


def test_eight_million_tets_on_one_gpu_against_the_oracle(beam38):
    """The cache-exceeding point of the bench once more, now WITH the oracle (round 4: an oracle step of 8.2M tets takes
    0.3 s on the GPU box's host with 16 threads, the operator 9.5 GB): K.d and 30 steps from a rough state, stepped as
    three block sets on three streams (split stepping, the default at this size) and with one launch of all blocks per
    step; 100 steps outside the suite: profiles/r04_long_parity_8M_tets.txt."""
    from oracle import fem_oracle as fo

    sol, lay, dt, l_M, F, (lmd, mu) = _build(beam38, 1, 0)
    st = sol.plan_stats()
    assert st["n_blocks"] == 2048 and st["threads"] == 512 and not sol.resident_kernel_info()["capable"]
    K = fo.MatrixFreeStiffness(lay.cells_local, beam38.points[lay.nodes], lmd, mu, threads=16)
    rng = np.random.default_rng(38)
    d = rng.uniform(-1e-2, 1e-2, size=(sol.n_dof, 1))
    assert rel_l2(sol.internal_force(d), K.dot(d)) < 1e-13
    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    steps = 30
    tn, o0, on = 0.25, d0, dn
    for _ in range(steps):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        on, o0, tn = o0, o1, tn + dt
    for split in (1, 0):
        sol.set_option("split_stepping", split)
        sol.set_state(d0, dn, 0.25)
        sol.step(steps)
        g0, gn, gt = sol.get_state()
        assert gt == tn and rel_l2(g0, o0) < 1e-11 and rel_l2(gn, on) < 1e-11, split
    sol.close()


@pytest.mark.parametrize("world,n,rank", [(2, 24, 1), (4, 30, 1)])
def test_rank_partitions_of_the_two_and_four_gpu_configurations(world, n, rank):
    """The per-GPU workloads of the driver's N = 2 and N = 4 scaling runs (bench.py: N_FOR_GPUS), which the 8-GPU test
    above does not cover: other block shapes - the 31-node cross-section of n = 30 cuts into 12 x 8 x 8-node boxes, whose
    blocks renumber their nodes with another axis running fastest (saa_plan.cpp: block_axis_order) - and other interface
    sizes.  The partial operator K_r against the oracle's and its free-body properties; 100 steps of the resident and the
    one-launch-per-step kernel against the oracle, exchange-free and through the peer exchange with loop-back (shared
    nodes then take 2 x their local force, which the oracle reproduces: Dynamic_solver.py:26-32)."""
    from oracle import fem_oracle as fo
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    steps = 100
    mesh = structured_beam(n)
    sol, lay, dt, l_M, F, (lmd, mu) = _build(mesh, world, rank)
    changed = sol.plan_stats()["n_renumbered"]
    assert (changed > 100) if n == 30 else (changed >= 0), changed
    assert sol.resident_kernel_info()["capable"] and len(lay.dirichlet_dofs) == 0
    pts = mesh.points[lay.nodes]
    del mesh
    K = fo.MatrixFreeStiffness(lay.cells_local, pts, lmd, mu)
    rng = np.random.default_rng(n)
    _operator_properties(sol, K, pts, rng)
    f1 = sol.internal_force(rng.uniform(-1e-2, 1e-2, size=(sol.n_dof, 1)))
    assert abs(f1.sum()) < 1e-9 * np.abs(f1).sum()  # a slab without clamped nodes is a free body

    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    sh_dof = lay.loc_dof_shared
    want = {}
    for mode in ("plain", "peer"):
        tn, o0, on = 0.25, d0, dn
        for _ in range(steps):
            f = K.dot(o0)
            if mode == "peer":
                f[sh_dof] = f[sh_dof] + f[sh_dof]  # this rank's force + the imaginary neighbour's copy of it, in rank order
            o1 = fo.cd_update(f, F, l_M, o0, on, dt, tn, ALPHA, lay.dirichlet_dofs)
            on, o0, tn = o0, o1, tn + dt
        want[mode] = (o0, on, tn)
    got = {}
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        sol.step(steps)
        got[name] = sol.get_state()
        assert got[name][2] == want["plain"][2]
        assert rel_l2(got[name][0], want["plain"][0]) < 1e-11 and rel_l2(got[name][1], want["plain"][1]) < 1e-11, name
    assert rel_l2(got["resident"][0], got["fused"][0]) < 1e-11 and np.abs(got["fused"][0]).max() > 1e-6
    sol.peer_attach_loopback(2)
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        sol.step_peer(steps)
        got[name] = sol.get_state()
        assert rel_l2(got[name][0], want["peer"][0]) < 1e-11 and rel_l2(got[name][1], want["peer"][1]) < 1e-11, name
    assert rel_l2(want["peer"][0], want["plain"][0]) > 1e-6  # (the doubled interface forces do change the field)
    sol.close()


def test_config4_predicted_window_on_the_middle_slab(middle_slab):
    """Config 4's per-GPU workload: the middle slab of the 8-GPU partition stepping through a sync-avoiding window - its
    3 042 shared nodes take the rows of a prediction table instead of being exchanged, and are recorded as history - in the
    resident kernel's predicted variant and in the one-launch-per-step kernel, against the oracle's restatement of the
    loop (Online_predictor.py:287-316: explicit step on the rank's own forces, overwrite of the shared dofs with the
    table row :298, history record :301)."""
    import torch
    from oracle import fem_oracle as fo

    ms = middle_slab
    lay, dt, l_M, F, K = ms["lay"], ms["dt"], ms["l_M"], ms["F"], ms["K"]
    sol = _slab_solver(ms)
    w, n_win = 3 * len(lay.shared_local), 64
    assert w == 9126 and sol.resident_kernel_info()["capable"]
    rng = np.random.default_rng(4)
    d0, dn = _rough_state(sol.n_dof, lay.dirichlet_dofs, rng)
    table_h = rng.uniform(-1e-4, 1e-4, size=(n_win + 8, w))
    table = torch.from_numpy(table_h).cuda()
    # oracle: 10 exchange-free steps (MODEL=True), the window, 9 more steps
    sh_dof = lay.loc_dof_shared
    tn, o0, on = 0.25, d0, dn
    hist_ref = np.zeros((n_win + 8, w))
    for i in range(10 + n_win + 9):
        o1 = fo.explicit_step(K, F, lay.dirichlet_dofs, tn, dt, o0, on, l_M, ALPHA)
        k = i - 10
        if 0 <= k < n_win:
            o1[sh_dof] = table_h[3 + k].reshape(-1, 1)       # Online_predictor.py:298
            hist_ref[5 + k] = o1[sh_dof, 0]                  # :301
        on, o0, tn = o0, o1, tn + dt
    got = {}
    for name, resident in (("resident", True), ("fused", False)):
        sol.set_resident_kernel(resident)
        sol.set_state(d0, dn, 0.25)
        hist = torch.zeros((n_win + 8, w), dtype=torch.float64, device="cuda")
        sol.step(10)                                  # synchronised steps would go here (exchange-free on one GPU)
        sol.step_predicted(n_win, table, 3, hist, 5)  # rows 3.. of the table, recorded from history row 5 on
        sol.step(9)
        torch.cuda.synchronize()
        assert torch.equal(hist[5:5 + n_win], table[3:3 + n_win]), name
        assert np.array_equal(hist.cpu().numpy(), hist_ref), name   # the oracle's history, bit for bit (copied values)
        got[name] = sol.get_state()
        assert got[name][2] == tn
        assert rel_l2(got[name][0], o0) < 1e-11 and rel_l2(got[name][1], on) < 1e-11, name
    assert rel_l2(got["resident"][0], got["fused"][0]) < 1e-11
    # the shared dofs ended the window on the last table row and then moved on with everybody else
    assert np.abs(got["fused"][0]).max() > 1e-6
    sol.close()
