#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING the unmodified reference.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

How the reference is driven (nothing of it is copied into this repo - its files are read from
``/root/reference`` at generation time and executed in place):

* ``Tools/*`` are imported as they are.  Three third-party modules they import at top level but do
  not need on this path are absent from the container (``meshio``, ``h5py``, ``mpi4py``;
  ``mgmetis`` likewise for the drivers), so ``sys.modules`` is pre-seeded with harness objects:
  an empty ``meshio`` (plus a no-op ``write_points_cells``), an ``h5py`` whose ``File`` keeps the
  ``Displacement`` dataset in an ``.npy`` file, and an ``mpi4py.MPI`` whose ``COMM_WORLD`` runs one
  *thread per rank* with barrier-based ``gather/bcast/Gather`` (so the reference's real ``syn_cpus``
  and the drivers' collectives execute with P=2).
* The four driver scripts are module-level code.  Their mesh-read + ParMETIS prologue cannot run
  (no ``meshio.read`` / ``part_mesh_kway``); everything after it is executed verbatim by
  ``exec``-ing the corresponding LINE RANGES of the reference files with the mesh arrays and an
  element->rank vector (geometric x-split; ParMETIS' own split cannot be reproduced - "partition
  parity unpinned") supplied in the namespace.  Hyper-parameters (``test_num`` etc.) are overridden
  to keep the fixtures small; each override is recorded in the fixture.

Outputs (all small ``.npz``): see the bottom of this file / tests/golden/README.md.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import threading
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

# ------------------------------------------------------------------------------------------
# harness modules
# ------------------------------------------------------------------------------------------
_tls = threading.local()


class ThreadComm:
    """COMM_WORLD look-alike: one Python thread per rank, rank kept in thread-local storage."""

    def __init__(self):
        self.size = 1
        self._barrier = threading.Barrier(1)
        self._slots = [None]
        self._box = None

    def configure(self, size):
        self.size = size
        self._barrier = threading.Barrier(size)
        self._slots = [None] * size

    def Get_rank(self):
        return getattr(_tls, "rank", 0)

    def Get_size(self):
        return self.size

    def gather(self, obj, root=0):
        r = self.Get_rank()
        self._slots[r] = obj
        self._barrier.wait()
        out = list(self._slots) if r == root else None
        self._barrier.wait()
        return out

    def bcast(self, obj, root=0):
        r = self.Get_rank()
        if r == root:
            self._box = obj
        self._barrier.wait()
        out = self._box
        self._barrier.wait()
        return out

    def Gather(self, sendbuf, recvbuf, root=0):
        parts = self.gather(np.array(sendbuf), root=root)
        if self.Get_rank() == root:
            recvbuf[...] = np.array(parts).reshape(recvbuf.shape)

    def Gatherv(self, sendbuf, recvbuf, root=0):
        parts = self.gather(np.array(sendbuf), root=root)
        if self.Get_rank() == root:
            recvbuf[...] = np.concatenate([np.atleast_1d(p) for p in parts])


COMM = ThreadComm()


class _H5File:
    """``h5py.File`` look-alike holding named datasets in ``<path>.npz``."""

    def __init__(self, path, mode="r"):
        self.path, self.mode, self.data = path, mode, {}
        if mode == "r":
            with np.load(path + ".npz") as z:
                self.data = {k: z[k] for k in z.files}

    def create_dataset(self, name, data=None, **_kw):
        self.data[name] = np.array(data)

    def __getitem__(self, name):
        return self.data[name]

    def close(self):
        if self.mode != "r":
            np.savez(self.path + ".npz", **self.data)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def install_harness():
    meshio = types.ModuleType("meshio")
    meshio.write_points_cells = lambda *a, **k: None
    h5py = types.ModuleType("h5py")
    h5py.File = _H5File
    mpi4py = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")
    MPI.COMM_WORLD = COMM
    mpi4py.MPI = MPI
    sys.modules.update({"meshio": meshio, "h5py": h5py, "mpi4py": mpi4py, "mpi4py.MPI": MPI})
    sys.path.insert(0, REF)


def ref_lines(fname, ranges, skip=()):
    """Source text of the given 1-based inclusive line ranges of a reference file (read, not copied)."""
    with open(os.path.join(REF, fname)) as fh:
        lines = fh.read().split("\n")
    out = []
    for lo, hi in ranges:
        for n in range(lo, hi + 1):
            out.append("" if n in skip else lines[n - 1])
    return "\n".join(out) + "\n"


def run_ranks(size, target):
    """Run ``target(rank)`` on ``size`` threads sharing COMM; returns the per-rank results."""
    COMM.configure(size)
    results, errors = [None] * size, []

    def body(r):
        _tls.rank = r
        try:
            results[r] = target(r)
        except BaseException as exc:  # noqa: BLE001
            errors.append(exc)
            COMM._barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results


class _RankStdout(io.TextIOBase):
    """Swallows the per-step prints of the drivers (rank 0 prints every step)."""

    def write(self, s):
        return len(s)


# ------------------------------------------------------------------------------------------
# golden runs
# ------------------------------------------------------------------------------------------
SNAP_SERIAL = (1, 10, 100, 1000, 5000, 10000)
SNAP_2RANK = (1, 10, 100, 1000, 5000)


def geometric_epart(points, cells, size):
    cent = points[cells, 0].mean(axis=1)
    order = np.argsort(cent, kind="stable")
    epart = np.zeros(len(cells), dtype=int)
    bounds = np.linspace(0, len(cells), size + 1).astype(int)
    for r in range(size):
        epart[order[bounds[r]:bounds[r + 1]]] = r
    return epart


def data_prepare(mesh, size, test_num, workdir):
    """Reference ``Data_prepare.py`` lines 1-50 (constants) and 104-246 on ``size`` thread-ranks."""
    prologue = ref_lines("Data_prepare.py", [(1, 50)], skip={5})  # line 5 = mgmetis import
    body = ref_lines("Data_prepare.py", [(104, 246)])
    epart = geometric_epart(mesh.points, mesh.tets, size)

    def rank_main(rank):
        g = {"__name__": "__ref_driver__"}
        exec(compile(prologue, "Data_prepare.py[1:50]", "exec"), g)
        g.update(test_num=test_num, Cells=mesh.tets.copy(), Facets=mesh.triangles.copy(),
                 Points=mesh.points.copy(), recvbuf=epart.copy(),
                 Mesh=types.SimpleNamespace(cells=None))
        exec(compile(body, "Data_prepare.py[104:246]", "exec"), g)
        keep = ("Local_ele_list Local_nodal_list shared_nodes Dirichlet_node Local_Dirichlet dt lumped_M "
                "F_pre d0 dn F_rankwise l_M d1_save LocalK").split()
        out = {k: g[k] for k in keep}
        out["Global_shared"] = g.get("Global_shared")
        return out

    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        with contextlib.redirect_stdout(_RankStdout()):
            res = run_ranks(size, rank_main)
    finally:
        os.chdir(cwd)
    return epart, res


def shared_extraction(size, workdir):
    src = ref_lines("Shared_extraction.py", [(1, 40)])

    def rank_main(rank):
        g = {"__name__": "__ref_driver__"}
        exec(compile(src, "Shared_extraction.py", "exec"), g)
        return {"shared_dof": np.array(g["shared_dof"]), "d": g["d"]}

    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        with contextlib.redirect_stdout(_RankStdout()):
            return run_ranks(size, rank_main)
    finally:
        os.chdir(cwd)


def online_predictor(mesh, size, epart, workdir, test_num, n_past, n_future, filter_size, hidden, seeds):
    """Reference ``Online_predictor.py`` lines 1-63 and 116-324 with seeded (untrained) models."""
    import torch
    from Tools.DNN_tools import LSTM_encoder_decoder

    prologue = ref_lines("Online_predictor.py", [(1, 63)], skip={7})  # line 7 = mgmetis import
    body = ref_lines("Online_predictor.py", [(116, 324)])
    weights = {}

    def rank_main(rank):
        g = {"__name__": "__ref_driver__"}
        exec(compile(prologue, "Online_predictor.py[1:63]", "exec"), g)
        g.update(test_num=test_num, n_past=n_past, n_future=n_future, filter_size=filter_size,
                 hidden_size=hidden, i_cri=n_past * filter_size - 1,
                 Cells=mesh.tets.copy(), Facets=mesh.triangles.copy(), Points=mesh.points.copy(),
                 recvbuf=epart.copy(), Mesh=types.SimpleNamespace(cells=None))
        # seeded state_dict at the path the driver builds (Online_predictor.py:139-140)
        lst = np.genfromtxt(f"Results/Shared_Data/Rank={rank}_shared.csv", delimiter=",")
        input_size = 3 * len(np.atleast_1d(lst))
        torch.manual_seed(seeds[rank])
        model = LSTM_encoder_decoder(input_size, g["hidden_size"], 2, True, 0.0, 0.0)
        mdir = (f"Distributed_save/Rank-{rank}/nB-{g['nB']}-nH-{g['hidden_size']}"
                f"-Lr-{g['learning_rate']}-filter={filter_size}")
        os.makedirs(mdir, exist_ok=True)
        torch.save(model.state_dict(), os.path.join(mdir, "model.pth"))
        weights[rank] = {k: v.numpy().copy() for k, v in model.state_dict().items()}
        exec(compile(body, "Online_predictor.py[116:324]", "exec"), g)
        keep = "d1_save d_sol_shared scale_max scale_min loc_dof_shared shared_nodes input_size dt".split()
        return {k: g[k] for k in keep}

    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        with contextlib.redirect_stdout(_RankStdout()):
            res = run_ranks(size, rank_main)
    finally:
        os.chdir(cwd)
    return res, weights


def main():
    install_harness()
    import torch
    from synchronization_avoiding_algorithms_amd.mesh import read_vtk

    import Tools.Mat_construction as MC
    import Tools.commons as CM
    import Tools.DNN_tools as DT
    import Tools.DNN_prediction as DP

    torch.set_num_threads(1)
    mesh = read_vtk(os.path.join(REF, "Mesh_info", "beam_coarse.vtk"))
    P, C, F = mesh.points, mesh.tets, mesh.triangles
    np.savez_compressed(os.path.join(HERE, "beam_coarse_mesh.npz"), points=P, tetra=C, triangle=F)

    E, nu, rho, fz = 1e6, 0.3, 1, 0.5
    lmd, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))

    # ---- 1. element operators (Local_K_coronary, Local_MKF) on 8 elements -------------------
    elas = CM.elasticity(lmd, mu, rho, fz, True)
    elas_steady = CM.elasticity(lmd, mu, rho, fz, False)
    sel = np.array([0, 1, 2, 37, 100, 128, 200, 255])
    Ke = np.array([MC.Local_K_coronary(1, 4, P[C[e]], elas) for e in sel])
    MKF = [MC.Local_MKF(1, 4, P[C[e]], elas_steady, None, None, 0) for e in sel]
    np.savez_compressed(os.path.join(HERE, "element_ops.npz"), elements=sel, coords=P[C[sel]], Ke=Ke,
                        Me=np.array([m[0] for m in MKF]), Ke_mkf=np.array([m[1] for m in MKF]),
                        Fe=np.array([m[2] for m in MKF]), lmd=lmd, mu=mu, rho=rho, fz=fz)
    print("element_ops.npz", Ke.shape)

    with tempfile.TemporaryDirectory() as tmp:
        # ---- 2. serial run: set-up vectors, K.d, trajectory snapshots -----------------------
        os.makedirs(os.path.join(tmp, "serial"))
        _, res = data_prepare(mesh, 1, max(SNAP_SERIAL), os.path.join(tmp, "serial"))
        r0 = res[0]
        rng = np.random.default_rng(0)
        d_rand = rng.uniform(-1e-2, 1e-2, size=(3 * len(P), 1))
        K = r0["LocalK"]
        np.savez_compressed(
            os.path.join(HERE, "serial_setup.npz"),
            local_nodes=np.array(r0["Local_nodal_list"]), local_elements=np.array(r0["Local_ele_list"]),
            dirichlet_nodes=np.array(r0["Dirichlet_node"]), local_dirichlet=np.array(r0["Local_Dirichlet"]),
            dt=r0["dt"], lumped_M=r0["lumped_M"], F_pre=r0["F_pre"], d0=r0["d0"], dn=r0["dn"],
            l_M=r0["l_M"], F_rankwise=r0["F_rankwise"],
            K_data=K.data, K_indices=K.indices, K_indptr=K.indptr, d_rand=d_rand, Kd_rand=K.dot(d_rand))
        snaps = {f"step_{s}": r0["d1_save"][:, s - 1] for s in SNAP_SERIAL}
        np.savez_compressed(os.path.join(HERE, "serial_trajectory.npz"), steps=np.array(SNAP_SERIAL), **snaps)
        print("serial: dt =", repr(float(r0["dt"])), "nnz =", K.nnz,
              "max|d| =", np.abs(r0["d1_save"][:, -1]).max())

        # ---- 3. two-rank run through the real syn_cpus -------------------------------------
        w2 = os.path.join(tmp, "two")
        os.makedirs(w2)
        epart, res2 = data_prepare(mesh, 2, max(SNAP_2RANK), w2)
        out = {"epart": epart, "dt": res2[0]["dt"], "Global_shared": np.array(res2[0]["Global_shared"]),
               "steps": np.array(SNAP_2RANK)}
        for r, rr in enumerate(res2):
            out[f"r{r}_local_nodes"] = np.array(rr["Local_nodal_list"])
            out[f"r{r}_local_elements"] = np.array(rr["Local_ele_list"])
            out[f"r{r}_shared_nodes"] = np.array(rr["shared_nodes"])
            out[f"r{r}_local_dirichlet"] = np.array(rr["Local_Dirichlet"])
            for s in SNAP_2RANK:
                out[f"r{r}_step_{s}"] = rr["d1_save"][:, s - 1]
        np.savez_compressed(os.path.join(HERE, "tworank_trajectory.npz"), **out)
        print("two-rank: nodes", [len(rr["Local_nodal_list"]) for rr in res2],
              "shared", [len(rr["shared_nodes"]) for rr in res2])

        # ---- 4. hybrid (sync-avoiding) loop, small hyper-parameters, untrained seeded models ---
        wh = os.path.join(tmp, "hyb")
        os.makedirs(wh)
        T, NP_, NF_, NS_, HID = 120, 4, 4, 5, 8
        epart, resd = data_prepare(mesh, 2, T, wh)
        ext = shared_extraction(2, wh)
        resh, weights = online_predictor(mesh, 2, epart, wh, T, NP_, NF_, NS_, HID, seeds=(11, 12))
        out = {"epart": epart, "test_num": T, "n_past": NP_, "n_future": NF_, "filter_size": NS_,
               "hidden_size": HID, "cut_off": 0.5, "dt": resh[0]["dt"]}
        for r in range(2):
            out[f"r{r}_shared_dof"] = ext[r]["shared_dof"]
            out[f"r{r}_shared_traj"] = ext[r]["d"]                 # Shared_extraction output
            out[f"r{r}_truth_last"] = resd[r]["d1_save"][:, -1]
            out[f"r{r}_scale"] = np.array([resh[r]["scale_max"], resh[r]["scale_min"]])
            out[f"r{r}_loc_dof_shared"] = np.array(resh[r]["loc_dof_shared"])
            out[f"r{r}_modeled"] = resh[r]["d1_save"]              # (ndof, T)
            out[f"r{r}_d_sol_shared"] = resh[r]["d_sol_shared"]
            for k, v in weights[r].items():
                out[f"r{r}_w::{k}"] = v
        np.savez_compressed(os.path.join(HERE, "hybrid_tworank.npz"), **out)
        print("hybrid: input sizes", [resh[r]["input_size"] for r in range(2)],
              "max|modeled|", [float(np.abs(resh[r]["d1_save"]).max()) for r in range(2)])

    # ---- 5. predictor table at reference-like shape -----------------------------------------
    n_p, n_f, n_s, in_sz, hid = 20, 20, 30, 24, 50
    torch.manual_seed(7)
    model = DT.LSTM_encoder_decoder(in_sz, hid, 2, True, 0.0, 0.0)
    n = n_p * n_s
    t = np.arange(n + 5)[:, None]
    j = np.arange(in_sz)[None, :]
    d_sol = np.round(1e-2 * np.sin(0.004 * t + 0.7 * j) * (1 + 0.1 * np.cos(0.001 * t * (j + 1))), 12)
    smax, smin = float(np.float32(d_sol.max() * 1.05)), float(np.float32(d_sol.min() * 1.05))
    with contextlib.redirect_stdout(_RankStdout()):
        NF = DP.encoder_decoder_predictor("cpu", n, model, n_p, n_f, n_s, in_sz, d_sol, smax, smin)
    out = {"n": n, "n_p": n_p, "n_f": n_f, "n_s": n_s, "input_size": in_sz, "hidden_size": hid,
           "d_sol": d_sol, "scale": np.array([smax, smin]), "NF": NF.astype(np.float32),
           "NF_is_fp32_exact": bool(np.all(NF.astype(np.float32).astype(np.float64) == NF))}
    for k, v in model.state_dict().items():
        out[f"w::{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "predictor_table.npz"), **out)
    print("predictor_table.npz", NF.shape, "fp32-exact:", out["NF_is_fp32_exact"])

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f:32s} {os.path.getsize(os.path.join(HERE, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
