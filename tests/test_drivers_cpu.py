"""Driver functions on CPU (solver double): artefact names / formats of the reference and the
Data_prepare semantics (which steps are saved), serial."""
import os

import numpy as np
import torch

from conftest import load_golden, rel_l2
from cpu_double import CpuSolverDouble, host_setup_fields
from synchronization_avoiding_algorithms_amd import drivers, results_io as rio


def test_data_prepare_serial_artefacts_and_trajectory(beam_coarse, tmp_path):
    g = load_golden("serial_trajectory.npz")
    setup = load_golden("serial_setup.npz")
    out = str(tmp_path)
    path, store = drivers.data_prepare(beam_coarse, 100, 1, out, 0, 1, tensor_device=torch.device("cpu"),
                                       solver_factory=lambda **kw: CpuSolverDouble(**kw), setup_fields=host_setup_fields)
    assert store.shape == (330, 100)
    for s in (1, 10, 100):  # column i holds the state after step i+1 (Data_prepare.py:238-240)
        assert rel_l2(store[:, s - 1], g[f"step_{s}"]) < 1e-13 if s > 1 else not store[:, 0].any()
    assert np.array_equal(rio.load_int_list(os.path.join(out, drivers.PATHS["local_nodes"].format(r=0))),
                          setup["local_nodes"])
    assert rio.load_int_list(os.path.join(out, drivers.PATHS["elements"].format(r=0))).tolist() == list(range(256))
    assert np.array_equal(rio.load_displacement(os.path.join(out, drivers.PATHS["truth"].format(r=0))), store)
    assert os.path.exists(os.path.join(out, "Results/Shared_Data/Global_shared.csv"))


def test_save_every_keeps_the_reference_columns(beam_coarse, tmp_path):
    full = drivers.data_prepare(beam_coarse, 23, 1, str(tmp_path / "a"), 0, 1, tensor_device=torch.device("cpu"),
                                solver_factory=lambda **kw: CpuSolverDouble(**kw), setup_fields=host_setup_fields)[1]
    thin = drivers.data_prepare(beam_coarse, 23, 5, str(tmp_path / "b"), 0, 1, tensor_device=torch.device("cpu"),
                                solver_factory=lambda **kw: CpuSolverDouble(**kw), setup_fields=host_setup_fields)[1]
    assert thin.shape == (330, 4)  # int(23/5) columns; steps i = 0, 5, 10, 15 (i % save_every == 0)
    assert np.array_equal(thin, full[:, [0, 5, 10, 15]])


def test_results_io_roundtrip(tmp_path):
    a = np.arange(12.0).reshape(3, 4)
    p = rio.save_displacement(str(tmp_path / "x" / "Local-rank-0.hdf5"), a)
    assert os.path.exists(p)
    assert np.array_equal(rio.load_displacement(str(tmp_path / "x" / "Local-rank-0.hdf5")), a)
    rio.save_int_list(str(tmp_path / "l.csv"), [4, 2, 9])
    assert rio.load_int_list(str(tmp_path / "l.csv")).tolist() == [4, 2, 9]
    rio.save_int_list(str(tmp_path / "one.csv"), [7])
    assert rio.load_int_list(str(tmp_path / "one.csv")).tolist() == [7]


def test_slab_partition_cuts_layered_meshes_on_node_planes():
    """The multi-GPU bench partitions (8 x-slabs, 7 interface planes x 1521 nodes at n=38):
    on a layered mesh every interface is ONE plane of nodes shared by exactly two parts."""
    from synchronization_avoiding_algorithms_amd import fem_setup as fs
    from synchronization_avoiding_algorithms_amd.mesh import clamp_nodes, slab_partition, structured_beam

    n, parts = 4, 3
    mesh = structured_beam(n)
    epart = slab_partition(mesh, parts)
    counts = np.bincount(epart, minlength=parts)
    assert counts.min() > 0 and counts.max() - counts.min() <= 6 * n * n  # at most one layer of cubes apart
    layouts, gshared = fs.build_layouts(mesh.tets, epart, parts, len(mesh.points), clamp_nodes(mesh))
    assert len(gshared) == (parts - 1) * (n + 1) ** 2
    assert [len(lay.shared_nodes) for lay in layouts] == [(n + 1) ** 2, 2 * (n + 1) ** 2, (n + 1) ** 2]
    for g in range(parts - 1):  # the nodes of one interface share one x coordinate
        left, right = set(layouts[g].nodes.tolist()), set(layouts[g + 1].nodes.tolist())
        assert len(np.unique(mesh.points[sorted(left & right), 0])) == 1
