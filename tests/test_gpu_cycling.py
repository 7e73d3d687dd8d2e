"""The cycling multi-step kernel (saa_hip.h: saa_multistep_kernel_info kind 2): one launch for many
steps of a partition that has several plan blocks per co-resident workgroup - the execution of
`parallel_explicit_solver_dis_pre` (Tools/Dynamic_solver.py:9-34) in the loop of Data_prepare.py:223-240 when the mesh is
too large for the resident kernel (8M tets on one GPU).

Small meshes cut into many tiny blocks make workgroups race through their blocks: a missing wait (read of a halo value
before its owner has written it, or overwrite of a buffer a neighbour still reads) shows up as a difference against the
one-launch-per-step execution of the same plan, whose arithmetic is identical (only the order of the LDS atomics differs:
1e-12), and against the oracle (1e-11, the tolerance of tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

from conftest import rel_l2
from test_gpu_parity import _oracle, _serial_solver

pytestmark = pytest.mark.gpu


def _rough_state(sol, seed):
    rng = np.random.default_rng(seed)
    d0 = rng.uniform(-1e-4, 1e-4, size=(sol.n_dof, 1))
    dn = d0 + rng.uniform(-1e-6, 1e-6, size=(sol.n_dof, 1))
    return d0, dn


@pytest.mark.parametrize("n,block_nodes,threads,grid", [(8, 20, 64, 64), (8, 20, 64, 8), (12, 64, 128, 128)])
def test_cycling_kernel_equals_one_launch_per_step(n, block_nodes, threads, grid, monkeypatch):
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(n)
    monkeypatch.setenv("SAA_NO_PERSISTENT", "1")  # plans this small fit the resident kernel, which has priority
    monkeypatch.setenv("SAA_NO_CYCLING", "1")
    fused, _, _, _, _ = _serial_solver(mesh, block_nodes=block_nodes, threads=threads)
    assert fused.multistep_kernel_info() == {"kind": "per_step", "workgroups": 0}
    monkeypatch.delenv("SAA_NO_CYCLING")
    monkeypatch.setenv("SAA_CYCLING_GRID", str(grid))
    cyc, lay, _, _, _ = _serial_solver(mesh, block_nodes=block_nodes, threads=threads)
    nb = cyc.plan_stats()["n_blocks"]
    assert cyc.multistep_kernel_info() == {"kind": "cycling", "workgroups": grid}
    assert nb % grid == 0 and nb // grid >= 2
    d0, dn = _rough_state(cyc, n)
    d0[lay.dirichlet_dofs] = 0
    dn[lay.dirichlet_dofs] = 0
    for sol in (fused, cyc):
        sol.set_state(d0, dn, 0.25)
    # mixed call lengths: buffer rotation between the per-step path (< 8 steps) and the cycling launches, 1000-step chunks
    for k in (1, 9, 2, 1001, 8, 1, 64):
        fused.step(k)
        cyc.step(k)
        a0, an, ta = fused.get_state()
        b0, bn, tb = cyc.get_state()
        assert ta == tb
        assert rel_l2(b0, a0) < 1e-12 and rel_l2(bn, an) < 1e-12, k
    # the handle can be taken off the multi-step kernels (saa_set_resident_kernel) and back
    cyc.set_resident_kernel(False)
    assert cyc.multistep_kernel_info()["kind"] == "per_step"
    cyc.set_resident_kernel(True)
    fused.step(40)
    cyc.step(40)
    assert rel_l2(cyc.get_state()[0], fused.get_state()[0]) < 1e-12
    fused.close()
    cyc.close()


def test_cycling_kernel_against_oracle(monkeypatch):
    fo = _oracle()
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(8)
    monkeypatch.setenv("SAA_NO_PERSISTENT", "1")
    monkeypatch.setenv("SAA_CYCLING_GRID", "32")
    sol, lay, dt, _, _ = _serial_solver(mesh, block_nodes=20, threads=64)
    assert sol.multistep_kernel_info()["kind"] == "cycling"
    ranks, odt, _, _ = fo.setup_problem(mesh.points, mesh.tets, mesh.triangles, 1, np.zeros(len(mesh.tets), dtype=int))
    assert odt == dt
    d0, dn = _rough_state(sol, 5)
    d0[ranks[0].dirichlet] = 0
    dn[ranks[0].dirichlet] = 0
    sol.set_state(d0, dn, 0.995)  # the load ramp ends inside the run (commons.py:7-11)
    tn, o0, on = 0.995, d0, dn
    for _ in range(300):
        o1 = fo.explicit_step(ranks[0].K, ranks[0].F, ranks[0].dirichlet, tn, dt, o0, on, ranks[0].l_M, 0.5)
        on, o0 = o0, o1
        tn = tn + dt
    sol.step(300)
    g0, gn, gt = sol.get_state()
    assert gt == tn and tn > 1.0
    assert rel_l2(g0, o0) < 1e-11 and rel_l2(gn, on) < 1e-11
    sol.close()


def test_recorder_keeps_the_per_step_path(monkeypatch):
    """The cycling kernel does not write trajectory columns: a handle with a recorder stays on one launch per step."""
    import torch
    from synchronization_avoiding_algorithms_amd.mesh import structured_beam

    mesh = structured_beam(8)
    monkeypatch.setenv("SAA_NO_PERSISTENT", "1")
    monkeypatch.setenv("SAA_CYCLING_GRID", "64")
    sol, lay, _, _, _ = _serial_solver(mesh, block_nodes=20, threads=64)
    ref, _, _, _, _ = _serial_solver(mesh, block_nodes=20, threads=64)
    d0, dn = _rough_state(sol, 1)
    for s in (sol, ref):
        s.set_state(d0, dn, 0.0)
    traj = torch.zeros((sol.n_dof, 30), dtype=torch.float64, device="cuda")
    sol.set_recorder(traj, save_every=1, next_step_index=0)
    sol.step(30)
    ref.step(30)  # cycling kernel
    torch.cuda.synchronize()
    assert rel_l2(traj[:, 29].cpu().numpy(), ref.get_state()[0].ravel()) < 1e-12
    sol.close()
    ref.close()
