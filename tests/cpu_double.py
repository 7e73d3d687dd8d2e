"""Test double of ``HipExplicitSolver`` for CPU-only (gloo) runs of the distributed orchestration.

Same methods and the same begin / all-reduce / finish contract as the C ABI (include/saa_hip.h), with
the numerics delegated to the CPU oracle.  Lives under tests/: the product never imports it."""
import numpy as np

from oracle import fem_oracle as fo
from synchronization_avoiding_algorithms_amd.fem_setup import host_setup_fields  # noqa: F401  (NumPy closed forms in
# place of the HIP set-up kernels: passed as PartitionedSolver(setup_fields=...) next to the solver double)


class CpuSolverDouble:
    def __init__(self, points, cells, l_M, F_rankwise, dirichlet_dofs, lmd, mu, dt, alpha, shared_local=(),
                 shared_slots=(), n_global_shared=0, device=0, block_nodes=0, threads=0, ramp=True):
        n = len(points)
        self.n_dof = 3 * n
        self.K = fo.assemble_local_stiffness(np.arange(n), np.asarray(cells, dtype=np.int64),
                                             np.asarray(points, dtype=np.float64), lmd, mu)
        self.F = np.asarray(F_rankwise, dtype=np.float64).reshape(-1, 1)
        self.m = np.asarray(l_M, dtype=np.float64).reshape(-1, 1)
        self.dirichlet = np.asarray(dirichlet_dofs, dtype=np.int64)
        self.dt, self.alpha = np.float64(dt), alpha
        self.sh_dof = fo.node_to_dof(np.asarray(shared_local, dtype=np.int64)) if len(shared_local) else \
            np.zeros(0, dtype=np.int64)
        self.slot_dof = fo.node_to_dof(np.asarray(shared_slots, dtype=np.int64)) if len(shared_slots) else \
            np.zeros(0, dtype=np.int64)
        mask = np.ones(3 * n_global_shared, dtype=bool)
        mask[self.slot_dof] = False
        self.foreign = np.nonzero(mask)[0]
        self.d0 = np.zeros((self.n_dof, 1))
        self.dn = np.zeros((self.n_dof, 1))
        self.tn = 0
        self.iface = None
        self._pending = None

    def set_stream(self, _ptr):
        pass

    def set_interface_buffer(self, iface):
        self.iface = iface

    def set_state(self, d0, dn, tn=0.0):
        self.d0 = np.array(d0, dtype=np.float64).reshape(-1, 1)
        self.dn = np.array(dn, dtype=np.float64).reshape(-1, 1)
        self.tn = tn

    def get_state(self):
        return self.d0.copy(), self.dn.copy(), self.tn

    def _rotate(self, d1):
        self.dn, self.d0 = self.d0, d1
        self.tn = self.tn + self.dt

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self._rotate(fo.cd_update(self.K.dot(self.d0), self.F, self.m, self.d0, self.dn, self.dt, self.tn,
                                      self.alpha, self.dirichlet))

    def step_begin(self):
        f = self.K.dot(self.d0)
        if self.iface is not None and len(self.slot_dof):
            self.iface.numpy()[self.slot_dof] = f[self.sh_dof, 0]
        self._pending = (f, fo.cd_update(f, self.F, self.m, self.d0, self.dn, self.dt, self.tn, self.alpha,
                                         self.dirichlet))

    def step_finish(self, hist=None, hist_row=0):
        f, d1 = self._pending
        self._pending = None
        if len(self.sh_dof):
            f = f.copy()
            f[self.sh_dof, 0] = self.iface.numpy()[self.slot_dof]
            full = fo.cd_update(f, self.F, self.m, self.d0, self.dn, self.dt, self.tn, self.alpha, self.dirichlet)
            d1[self.sh_dof] = full[self.sh_dof]
            if hist is not None:
                hist.numpy()[hist_row, :] = d1[self.sh_dof, 0]
        if self.iface is not None:
            self.iface.numpy()[self.foreign] = 0.0
        self._rotate(d1)

    def step_predicted(self, nsteps, table, table_row0=0, hist=None, hist_row0=0):
        for k in range(nsteps):
            d1 = fo.cd_update(self.K.dot(self.d0), self.F, self.m, self.d0, self.dn, self.dt, self.tn, self.alpha,
                              self.dirichlet)
            if len(self.sh_dof):
                d1[self.sh_dof, 0] = table.numpy()[table_row0 + k, :]
                if hist is not None:
                    hist.numpy()[hist_row0 + k, :] = d1[self.sh_dof, 0]
            self._rotate(d1)

    def close(self):
        pass
