"""One mesh partition per GPU: the synchronised / sync-avoiding stepping of the reference drivers.

Reference behaviour being replaced (all in /root/reference):

* ``syn_cpus`` (``Tools/Distributed_tools.py:77-92``): every step each rank pickles its whole local
  force vector *and* its node list to rank 0, which scatter-adds into a global ``(3N,1)`` vector and
  broadcasts it back.  Here only the forces of *shared* nodes travel: they are written by the fused
  step kernel into a compact interface buffer indexed by the sorted ``Global_shared`` list
  (``Data_prepare.py:121-124``), summed with ONE ``all_reduce`` (RCCL over xGMI when the process
  group is ``nccl``) and applied to the shared nodes by a small finish kernel.  Interior nodes never
  wait for the collective: their update is already done when it starts.
* the hybrid loop of ``Online_predictor.py:251-318``: ``n_past*filter_size`` synchronised steps, then
  windows of ``n_future*filter_size`` steps without any communication in which the shared dofs are
  overwritten by LSTM predictions (``:298``) and recorded as the next window's history (``:301``).

The per-partition solver is injected (``solver_factory``) so that the orchestration in this file is
exercised on CPU with ``gloo`` by the tests; the product factory is :class:`HipExplicitSolver`.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from . import fem_setup as fs
from .mesh import clamp_nodes


def _hip_factory(**kw):
    from .solver import HipExplicitSolver

    return HipExplicitSolver(**kw)


class PartitionedSolver:
    """This rank's partition of a replicated mesh (``Data_prepare.py:76-79``) plus its exchange."""

    def __init__(self, points, cells, facets_or_dirichlet_nodes, epart, rank, world,
                 E=1e6, nu=0.3, rho=1.0, fz=0.5, alpha=0.5, gamma=0.9, device=0, process_group=None,
                 tensor_device=None, solver_factory: Optional[Callable] = None, block_nodes=0, threads=0,
                 native_exchange=True, exchange="auto", setup_fields: Optional[Callable] = None,
                 wait_timeout_s: Optional[float] = None, resident_on_shared_device=False):
        import torch

        self.rank, self.world = int(rank), int(world)
        self.group = process_group
        points = np.ascontiguousarray(points, dtype=np.float64)
        cells = np.asarray(cells, dtype=np.int64)
        dn, facets = np.asarray(facets_or_dirichlet_nodes), None
        if dn.ndim == 2:  # triangle facets: detect the clamp like Data_prepare.py:127-136
            if setup_fields is None:  # on the device, with the rest of the layout (saa_topology_build)
                dn, facets = None, dn
            else:
                from .mesh import Mesh

                dn = clamp_nodes(Mesh(points, {"tetra": cells, "triangle": dn}))
        self.lmd, self.mu = fs.lame(E, nu)
        # this rank's layout only, from the topology kernels; lumped mass / load / shortest edge from the HIP set-up
        # kernels on the elements that touch this rank's nodes (``setup_fields`` is injectable for the CPU-only tests of
        # this orchestration, which then take the NumPy layout as well)
        self.layout, self.global_shared, l_M, F_rankwise, dt_local = fs.rank_problem(
            points, cells, dn, epart, rank, world, E, nu, rho, fz, gamma, device, setup_fields, facets=facets)
        lay = self.layout
        # dt = min over the ranks of the local CFL steps (Data_prepare.py:147-154)
        self.dt = dt_local
        if self.world > 1:
            import torch.distributed as dist

            box = [None] * self.world
            dist.all_gather_object(box, float(dt_local), group=process_group)
            self.dt = min(box)
        self.alpha = alpha
        factory = solver_factory or _hip_factory
        self.solver = factory(points=points[lay.nodes], cells=lay.cells_local, l_M=l_M,
                              F_rankwise=F_rankwise, dirichlet_dofs=lay.dirichlet_dofs,
                              lmd=self.lmd, mu=self.mu, dt=self.dt, alpha=alpha,
                              shared_local=lay.shared_local, shared_slots=lay.shared_slots,
                              n_global_shared=len(self.global_shared), device=device,
                              block_nodes=block_nodes, threads=threads)
        self.tensor_device = tensor_device if tensor_device is not None else torch.device("cuda", device)
        self.iface = torch.zeros(3 * len(self.global_shared), dtype=torch.float64, device=self.tensor_device)
        if self.world > 1:
            self.solver.set_interface_buffer(self.iface)
        if self.tensor_device.type == "cuda":
            self.solver.set_stream(torch.cuda.current_stream(self.tensor_device).cuda_stream)
        self.input_size = 3 * len(lay.shared_nodes)          # Online_predictor.py:126
        self.steps_done = 0
        self.device_ordinal = int(device)
        # ranks sharing one GPU (rehearsals, the one-GPU test box) keep one launch per step unless told otherwise: a
        # resident kernel waiting for another process' kernel only advances by time-slicing
        self._resident_on_shared_device = bool(resident_on_shared_device)
        if wait_timeout_s is not None and hasattr(self.solver, "set_option"):
            self.solver.set_option("wait_timeout_s", wait_timeout_s)  # bound of the in-kernel waits for other ranks
        # how the shared-node forces travel in synchronised steps (all ranks agree on one of them):
        #   "peer"  direct xGMI peer stores + rank-ordered sums (saa_step_peer), no collective
        #   "rccl"  ncclAllReduce issued from C++ (saa_step_synced)
        #   "torch" torch.distributed.all_reduce between saa_step_begin / saa_step_finish
        if exchange not in ("auto", "peer", "rccl", "torch"):
            raise ValueError("exchange must be auto, peer, rccl or torch")
        self.exchange = "torch"
        if native_exchange and exchange != "torch" and self.world > 1 and self.tensor_device.type == "cuda":
            if exchange in ("auto", "peer") and self._init_peer_exchange():
                self.exchange = "peer"
            elif exchange in ("auto", "rccl") and self._init_native_exchange():
                self.exchange = "rccl"
        self.native_exchange = self.exchange == "rccl"

    def _agree(self, ok) -> bool:
        """True iff every rank reports success (object all-gather: works on nccl and gloo groups)."""
        import torch.distributed as dist

        box = [None] * self.world
        dist.all_gather_object(box, int(bool(ok)), group=self.group)
        return min(box) == 1

    def _init_peer_exchange(self) -> bool:
        """Map every neighbour's inbox through HIP IPC (``saa_peer_export`` / ``saa_peer_attach``) and prove the
        path with one exchange of known values; on any failure on any rank all ranks keep the all-reduce."""
        import torch.distributed as dist

        if not hasattr(self.solver, "peer_export") or self.world > 64:
            return self._agree(False) and False
        slots = np.asarray(self.layout.shared_slots, dtype=np.int32)
        ok, handle, order = 1, bytes(64), np.zeros(len(slots), dtype=np.int32)
        try:
            handle, order = self.solver.peer_export(self.world)
        except Exception:  # noqa: BLE001
            ok = 0
        info = (ok, handle, self.device_ordinal, slots, order, self._physical_device_id())
        box = [None] * self.world
        dist.all_gather_object(box, info, group=self.group)
        if (len({b[5] for b in box}) < self.world and hasattr(self.solver, "set_resident_kernel")
                and not self._resident_on_shared_device):
            self.solver.set_resident_kernel(False)
        if min(b[0] for b in box) == 0:
            return False
        try:
            self.solver.peer_attach(self.rank, self.world, [b[1] for b in box], [b[2] for b in box],
                                    [b[3] for b in box], [b[4] for b in box])
        except Exception:  # noqa: BLE001
            ok = 0
        if not self._agree(ok):  # also the barrier: every inbox is mapped before anybody pushes
            return False
        try:
            ok = int(self.solver.peer_selftest())
        except Exception:  # noqa: BLE001
            ok = 0
        return self._agree(ok)

    def _physical_device_id(self):
        """Something that tells two physical GPUs apart even when a launcher shows every rank its own GPU as ordinal 0
        (HIP_VISIBLE_DEVICES): UUID, else PCI address, else host name + ordinal."""
        import socket

        import torch

        try:
            p = torch.cuda.get_device_properties(self.device_ordinal)
            for name in ("uuid", "pci_bus_id"):
                v = getattr(p, name, None)
                if v is not None and str(v) not in ("", "None"):
                    extra = (getattr(p, "pci_domain_id", 0), getattr(p, "pci_device_id", 0)) if name == "pci_bus_id" else ()
                    return (socket.gethostname(), name, str(v), *extra)
        except Exception:  # noqa: BLE001
            pass
        return (socket.gethostname(), "ordinal", self.device_ordinal)

    def _init_native_exchange(self) -> bool:
        """Join an RCCL communicator owned by the C++ side (``saa_comm_init``); on any failure keep the
        ``torch.distributed`` all-reduce.  All ranks take the same decision."""
        import torch
        import torch.distributed as dist

        ok, uid = 1, bytes(128)
        try:
            if dist.get_backend(self.group) != "nccl" or not hasattr(self.solver, "comm_init"):
                ok = 0
            elif self.rank == 0:
                uid = self.solver.comm_unique_id()
        except Exception:  # noqa: BLE001
            ok = 0
        box = [uid]
        dist.broadcast_object_list(box, src=0, group=self.group)
        flag = torch.tensor([ok], device=self.tensor_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            return False
        try:
            self.solver.comm_init(box[0], self.rank, self.world)
        except Exception:  # noqa: BLE001
            ok = 0
        flag = torch.tensor([ok], device=self.tensor_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    # -- the three kinds of step ---------------------------------------------------------------------
    def step_synced(self, nsteps=1, hist=None, hist_row0=0):
        """``MODEL=False`` steps (``Dynamic_solver.py:22-32``): local update, all-reduce of the shared
        forces, shared-node update; optional history record (``Online_predictor.py:260``)."""
        import torch.distributed as dist

        if self.world == 1:  # serial: nothing is shared, nothing to exchange or record
            self.solver.step(nsteps)
            self.steps_done += nsteps
            return
        if self.exchange == "peer":  # fused kernel + exchange kernel per step, no collective
            self.solver.step_peer(nsteps, hist, hist_row0)
        elif self.exchange == "rccl":  # fused kernel, ncclAllReduce and finish kernel enqueued from C++
            self.solver.step_synced(nsteps, hist, hist_row0)
        else:
            for k in range(nsteps):
                self.solver.step_begin()
                dist.all_reduce(self.iface, group=self.group)
                self.solver.step_finish(hist, hist_row0 + k)
        self.steps_done += nsteps

    def step_local(self, nsteps=1):
        """``MODEL=True`` steps without overwrite: every rank advances on its own partial forces."""
        self.solver.step(nsteps)
        self.steps_done += nsteps

    def step_predicted(self, nsteps, table, table_row0=0, hist=None, hist_row0=0):
        """Sync-free steps with the halo overwrite of ``Online_predictor.py:287-316``."""
        self.solver.step_predicted(nsteps, table, table_row0, hist, hist_row0)
        self.steps_done += nsteps

    def reconcile_shared(self):
        """COLLECTIVE.  Every holder's copy of a shared node's d^n and d^(n-1) is replaced by the mean over its holders.
        In predicted steps every rank writes ITS model's values into its copy (``Online_predictor.py:298``), so the
        copies differ; synchronised steps sum the forces but update each copy from itself, so the copies would stay apart
        (and keep a difference in velocity).  Not in the reference, which never returns to synchronised steps."""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return
        d0, dn, tn = self.solver.get_state()
        loc = np.asarray(self.layout.loc_dof_shared, dtype=np.int64)
        gd = (3 * np.asarray(self.layout.shared_slots, dtype=np.int64)[:, None] + np.arange(3)[None, :]).ravel()
        buf = np.zeros((3, 3 * len(self.global_shared)))
        buf[0, gd], buf[1, gd], buf[2, gd] = d0[loc, 0], dn[loc, 0], 1.0
        t = torch.from_numpy(buf).to(self.tensor_device)
        dist.all_reduce(t, group=self.group)  # (every rank receives the same sums: the copies end up bit-identical)
        buf = t.cpu().numpy()
        d0[loc, 0], dn[loc, 0] = buf[0, gd] / buf[2, gd], buf[1, gd] / buf[2, gd]
        self.solver.set_state(d0, dn, tn)

    # -- convenience ---------------------------------------------------------------------------------
    def get_state(self):
        return self.solver.get_state()

    def close(self):
        self.solver.close()


def run_hybrid(part: PartitionedSolver, n_steps, predictor, n_past, n_future, filter_size, save=None, resync_every=None,
               resync_steps=None):
    """``Online_predictor.py:251-318`` on one rank.

    ``predictor(n, hist) -> table``: ``hist`` is the ``(n_steps, input_size)`` float64 history tensor on the
    solver's device (``d_sol_shared``), the result a ``(n_future*filter_size, input_size)`` float64 tensor
    (``encoder_decoder_predictor``).  ``save(i, part)`` is called after every step that the reference would
    store (``save_every`` logic lives in the callback).  Returns the history tensor.

    ``resync_every=k`` is an extension, not reference behaviour (the reference never synchronises again after the warm-up;
    BASELINE.json's configs[4] asks for "RCCL every k-th step only"): after every ``k`` predicted windows the next
    ``resync_steps`` steps (default: one window, ``n_future*filter_size``) exchange the shared-node forces like the warm-up
    does and record their true values in the history, from which the following windows are predicted; they start from the
    mean of the holders' copies of every shared node (:meth:`PartitionedSolver.reconcile_shared`).  Every rank must pass
    the same values: synchronised steps are collective.  Measured on the reference's example it does not pay - the field
    gets worse inside every synchronised window and with ``k = 1`` the loop diverges (DESIGN.md section 5,
    ``profiles/r03_resync_every_k.txt``): off by default.
    """
    if resync_every is not None and int(resync_every) < 1:
        raise ValueError("resync_every must be a positive number of windows (or None)")
    if resync_steps is None:
        resync_steps = n_future * filter_size
    if int(resync_steps) < 1:
        raise ValueError("resync_steps must be positive")
    import torch

    hist = torch.zeros((n_steps, max(part.input_size, 1)), dtype=torch.float64, device=part.tensor_device)
    hist = hist[:, :part.input_size] if part.input_size else hist[:, :0]
    hist = hist.contiguous()
    i_cri = n_past * filter_size - 1
    window = n_future * filter_size
    i, resync_left, windows = 0, 0, 0
    while i < n_steps:
        if i <= i_cri or resync_left > 0:
            warm_up = i <= i_cri  # (no window has run yet: resync_left is 0)
            todo = i_cri + 1 - i if warm_up else resync_left
            n = min(todo, n_steps - i) if save is None else 1
            part.step_synced(n, hist if part.input_size else None, i)
            i += n
            if not warm_up:
                resync_left -= n
            if save is not None:
                save(i - 1, part)
        else:
            table = predictor(i, hist)
            todo = min(window, n_steps - i)
            if save is None:
                part.step_predicted(todo, table, 0, hist if part.input_size else None, i)
                i += todo
            else:
                for k in range(todo):
                    part.step_predicted(1, table, k, hist if part.input_size else None, i)
                    i += 1
                    save(i - 1, part)
            windows += 1
            if resync_every and windows % int(resync_every) == 0 and i < n_steps:
                resync_left = int(resync_steps)
                part.reconcile_shared()
    return hist
