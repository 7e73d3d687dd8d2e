"""Device-resident explicit solver: Python face of one ``saa_solver`` handle of ``libsaa_hip.so``.

``HipExplicitSolver`` owns the state ``(d0, dn, tn)`` of ``Time_integration_displacement``
(/root/reference ``Tools/commons.py:47-55``) on the GPU and advances it with the fused HIP kernel;
arrays cross this boundary in the caller's (rank-local, first-touch) numbering only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _f64(a, n=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a.size else None


def _dev(t):
    """Device pointer of a torch CUDA tensor (or None)."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 8):
        raise ValueError("device buffers must be contiguous float64 CUDA tensors")
    return C.c_void_p(t.data_ptr())


class HipExplicitSolver:
    """One mesh partition on one MI355X.

    Parameters follow the variables of ``Data_prepare.py:200-209``:
    ``points`` = ``Points[Local_nodal_list]`` (n,3); ``cells`` local node ids (ne,4);
    ``l_M`` / ``F_rankwise`` (3n,) or (3n,1); ``dirichlet_dofs`` = ``Local_Dirichlet``.
    ``shared_local`` / ``shared_slots`` describe this rank's interface nodes
    (:class:`fem_setup.RankLayout`), ``n_global_shared`` = ``len(Global_shared)``.
    """

    def __init__(self, points, cells, l_M, F_rankwise, dirichlet_dofs, lmd, mu, dt, alpha,
                 shared_local=(), shared_slots=(), n_global_shared=0, ramp=True, device=0,
                 block_nodes=0, threads=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        pts = _f64(points)
        self.n_nodes = pts.size // 3
        self.n_dof = 3 * self.n_nodes
        tets = _i32(cells)
        self.n_elems = tets.size // 4
        mass, fext = _f64(l_M, self.n_dof), _f64(F_rankwise, self.n_dof)
        dd, sn, ss = _i32(dirichlet_dofs), _i32(shared_local), _i32(shared_slots)
        if sn.size != ss.size:
            raise ValueError("shared_local and shared_slots differ in length")
        self.n_shared = int(sn.size)
        self.n_global_shared = int(n_global_shared)
        self.dt = float(dt)
        self.device = int(device)
        pb = _lib.Problem(
            n_nodes=self.n_nodes, n_elems=self.n_elems, xyz=_dptr(pts), tets=_iptr(tets),
            lumped_mass=_dptr(mass), f_ext=_dptr(fext), dirichlet_dofs=_iptr(dd), n_dirichlet=dd.size,
            shared_nodes=_iptr(sn), shared_slots=_iptr(ss), n_shared=sn.size,
            n_global_shared=self.n_global_shared, lambda_=float(lmd), mu=float(mu), dt=float(dt),
            alpha=float(alpha), ramp=1 if ramp else 0, device=int(device), block_nodes=int(block_nodes),
            threads=int(threads))
        _lib.check(self._lib.saa_create(C.byref(pb), C.byref(self._h)))
        self._iface = None

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.saa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing -------------------------------------------------------------------------------
    def plan_stats(self) -> dict:
        st = _lib.PlanStats()
        _lib.check(self._lib.saa_plan_stats_get(self._h, C.byref(st)))
        return st.as_dict()

    def resident_kernel_info(self) -> dict:
        """Whether multi-step calls run through the resident kernel, its LDS footprint and steps per launch."""
        cap, lds, spl = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._lib.saa_resident_kernel_info(self._h, C.byref(cap), C.byref(lds), C.byref(spl)))
        return {"capable": bool(cap.value), "lds_bytes": lds.value, "steps_per_launch": spl.value}

    def set_recorder(self, traj=None, save_every=1, next_step_index=0):
        """``traj``: float64 CUDA tensor ``(3*n_nodes, n_cols)`` (kept alive by the caller) that receives the
        displacement of every ``save_every``-th step, like ``d1_save`` of ``Data_prepare.py:236-240``; ``None``
        switches recording off."""
        if traj is None:
            _lib.check(self._lib.saa_set_recorder(self._h, None, 0, 1, 0))
            self._traj = None
            return
        if traj.dim() != 2 or traj.shape[0] != self.n_dof:
            raise ValueError(f"trajectory matrix must be ({self.n_dof}, n_cols)")
        _lib.check(self._lib.saa_set_recorder(self._h, _dev(traj), int(traj.shape[1]), int(save_every),
                                              int(next_step_index)))
        self._traj = traj

    def set_resident_kernel(self, enable: bool):
        _lib.check(self._lib.saa_set_resident_kernel(self._h, 1 if enable else 0))

    def set_option(self, name: str, value: float):
        """Run-time options by name (``saa_set_option``): ``synced_graph`` (1/0: graph replays / eager launches in
        :meth:`step_synced`), ``wait_timeout_s`` (bound of every in-kernel wait for another workgroup or rank)."""
        _lib.check(self._lib.saa_set_option(self._h, name.encode(), float(value)))

    def set_deterministic(self, enable: bool):
        """Atomic-free two-kernel steps with a fixed summation order: bit-identical results from run to run
        (``saa_set_deterministic``; a verification mode, several times slower)."""
        _lib.check(self._lib.saa_set_deterministic(self._h, 1 if enable else 0))

    def set_stream(self, stream_ptr):
        """``stream_ptr``: integer hipStream_t, e.g. ``torch.cuda.current_stream().cuda_stream``."""
        _lib.check(self._lib.saa_set_stream(self._h, C.c_void_p(int(stream_ptr) if stream_ptr else 0)))

    def synchronize(self):
        _lib.check(self._lib.saa_synchronize(self._h))

    # -- state ----------------------------------------------------------------------------------
    def set_state(self, d0, dn, tn=0.0):
        a, b = _f64(d0, self.n_dof), _f64(dn, self.n_dof)
        _lib.check(self._lib.saa_set_state(self._h, _dptr(a), _dptr(b), float(tn)))

    def get_state(self):
        """Returns ``(d0, dn, tn)`` with the vectors shaped ``(3n,1)`` like the reference's."""
        d0, dn, tn = np.empty(self.n_dof), np.empty(self.n_dof), C.c_double()
        _lib.check(self._lib.saa_get_state(self._h, _dptr(d0), _dptr(dn), C.byref(tn)))
        return d0.reshape(-1, 1), dn.reshape(-1, 1), tn.value

    def get_state_device(self, d0_out=None, dn_out=None):
        _lib.check(self._lib.saa_get_state_device(self._h, _dev(d0_out), _dev(dn_out)))

    def set_loads(self, F_rankwise=None, l_M=None):
        f = _f64(F_rankwise, self.n_dof) if F_rankwise is not None else None
        m = _f64(l_M, self.n_dof) if l_M is not None else None
        _lib.check(self._lib.saa_set_loads(self._h, _dptr(f) if f is not None else None,
                                           _dptr(m) if m is not None else None))

    # -- operators ------------------------------------------------------------------------------
    def internal_force(self, d):
        """``LocalK.dot(d)`` (``Dynamic_solver.py:12``), matrix-free on the GPU -> ``(3n,1)``."""
        a, f = _f64(d, self.n_dof), np.empty(self.n_dof)
        _lib.check(self._lib.saa_internal_force(self._h, _dptr(a), _dptr(f)))
        return f.reshape(-1, 1)

    def internal_force_device(self, d, out):
        """``out = K_local . d`` for float64 CUDA tensors of ``3*n_nodes`` values (caller numbering)."""
        if d.numel() != self.n_dof or out.numel() != self.n_dof:
            raise ValueError(f"expected {self.n_dof} values")
        _lib.check(self._lib.saa_internal_force_device(self._h, _dev(d), _dev(out)))
        return out

    def cd_update(self, f_int, d0, dn, tn):
        a, b, c = _f64(f_int, self.n_dof), _f64(d0, self.n_dof), _f64(dn, self.n_dof)
        out = np.empty(self.n_dof)
        _lib.check(self._lib.saa_cd_update(self._h, _dptr(a), _dptr(b), _dptr(c), float(tn), _dptr(out)))
        return out.reshape(-1, 1)

    # -- stepping -------------------------------------------------------------------------------
    def step(self, nsteps=1):
        _lib.check(self._lib.saa_step(self._h, int(nsteps)))

    def set_interface_buffer(self, iface):
        """``iface``: float64 CUDA tensor of ``3*n_global_shared`` zeros, kept alive by the caller."""
        if iface is not None and iface.numel() != 3 * self.n_global_shared:
            raise ValueError("interface buffer must hold 3*n_global_shared doubles")
        self._iface = iface
        _lib.check(self._lib.saa_set_interface_buffer(self._h, _dev(iface)))

    def step_begin(self):
        _lib.check(self._lib.saa_step_begin(self._h))

    def step_finish(self, hist=None, hist_row=0):
        _lib.check(self._lib.saa_step_finish(self._h, _dev(hist), int(hist_row)))

    # -- native exchange (RCCL from C++) ----------------------------------------------------------
    @staticmethod
    def rccl_library_path():
        """The RCCL build PyTorch-ROCm already has in the process (one copy of the library only)."""
        import os

        import torch

        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return path if os.path.exists(path) else "librccl.so"

    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        _lib.check(self._lib.saa_comm_unique_id(self.rccl_library_path().encode(), buf))
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self._lib.saa_comm_init(self._h, self.rccl_library_path().encode(), buf, int(rank), int(world)))

    def step_synced(self, nsteps=1, hist=None, hist_row0=0):
        """``nsteps`` synchronised steps entirely enqueued from C++ (needs :meth:`comm_init`)."""
        _lib.check(self._lib.saa_step_synced(self._h, int(nsteps), _dev(hist), int(hist_row0)))

    # -- direct peer exchange (xGMI peer stores, no collective) -------------------------------------
    def peer_export(self, world: int):
        """Returns ``(handle, order)``: the 64-byte IPC handle of this rank's inbox and, per shared node, its
        position in this rank's push order; both are all-gathered by the caller."""
        buf = (C.c_uint8 * 64)()
        order = np.zeros(max(self.n_shared, 1), dtype=np.int32)
        _lib.check(self._lib.saa_peer_export(self._h, int(world), buf, order.ctypes.data_as(C.POINTER(C.c_int32))))
        return bytes(buf), order[:self.n_shared]

    def peer_attach(self, rank: int, world: int, handles, devices, slot_lists, order_lists):
        """``handles``: per-rank 64-byte IPC handles; ``devices``: per-rank HIP ordinals; ``slot_lists`` /
        ``order_lists``: per-rank ``shared_slots`` and push orders (all as all-gathered by the caller)."""
        def cat(lists):
            n = sum(len(x) for x in lists)
            return _i32(np.concatenate([np.asarray(x, dtype=np.int32).reshape(-1) for x in lists]) if n
                        else np.zeros(1, dtype=np.int32))

        if [len(x) for x in slot_lists] != [len(x) for x in order_lists]:
            raise ValueError("slot and order lists differ in length")
        hb = (C.c_uint8 * (64 * world)).from_buffer_copy(b"".join(handles))
        dev, counts = _i32(devices), _i32([len(x) for x in slot_lists])
        slots, orders = cat(slot_lists), cat(order_lists)
        p32 = C.POINTER(C.c_int32)
        _lib.check(self._lib.saa_peer_attach(self._h, int(rank), int(world), hb, dev.ctypes.data_as(p32),
                                             counts.ctypes.data_as(p32), slots.ctypes.data_as(p32),
                                             orders.ctypes.data_as(p32)))

    def peer_attach_loopback(self, world: int = 2):
        """One-GPU rehearsal of the peer exchange (``saa_peer_attach_loopback``): ``world - 1`` imaginary neighbours
        holding this rank's shared nodes; shared nodes are then updated with ``world`` x their local partial force."""
        _lib.check(self._lib.saa_peer_attach_loopback(self._h, int(world)))

    def peer_selftest(self) -> bool:
        ok = C.c_int32()
        _lib.check(self._lib.saa_peer_selftest(self._h, C.byref(ok)))
        return bool(ok.value)

    def step_peer(self, nsteps=1, hist=None, hist_row0=0):
        """``nsteps`` synchronised steps through the peer exchange (needs :meth:`peer_attach`)."""
        _lib.check(self._lib.saa_step_peer(self._h, int(nsteps), _dev(hist), int(hist_row0)))

    def step_predicted(self, nsteps, table, table_row0=0, hist=None, hist_row0=0):
        _lib.check(self._lib.saa_step_predicted(self._h, int(nsteps), _dev(table), int(table_row0),
                                                _dev(hist), int(hist_row0)))

    def halo_gather(self, row):
        _lib.check(self._lib.saa_halo_gather(self._h, _dev(row)))

    def halo_scatter(self, row):
        _lib.check(self._lib.saa_halo_scatter(self._h, _dev(row)))

    def time_steps(self, nsteps) -> float:
        """Milliseconds (HIP events on the solver's stream) for ``nsteps`` exchange-free steps."""
        ms = C.c_double()
        _lib.check(self._lib.saa_time_steps(self._h, int(nsteps), C.byref(ms)))
        return ms.value


def plan_host_stats(points, cells, block_nodes=0) -> dict:
    """Block-plan statistics without touching a GPU (``saa_plan_host_stats``)."""
    lib = _lib.load()
    pts, tets = _f64(points), _i32(cells)
    st = _lib.PlanStats()
    _lib.check(lib.saa_plan_host_stats(pts.size // 3, tets.size // 4, _dptr(pts), _iptr(tets),
                                       int(block_nodes), C.byref(st)))
    return st.as_dict()


def plan_host_check(points, cells, block_nodes=0) -> int:
    """Number of violated invariants of the block plan the library builds for this partition (``saa_plan_host_check``: the
    numbering is a permutation, every item's tets are mesh elements in the mesh's orientation, every element exactly once in
    every block owning one of its nodes); 0 = sound.  No GPU needed."""
    lib = _lib.load()
    pts, tets = _f64(points), _i32(cells)
    bad = C.c_int64()
    _lib.check(lib.saa_plan_host_check(pts.size // 3, tets.size // 4, _dptr(pts), _iptr(tets), int(block_nodes), C.byref(bad)))
    return int(bad.value)
