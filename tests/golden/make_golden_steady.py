#!/usr/bin/env python3
"""Golden fixture of the reference's steady solve (Data_prepare.py:157-168): RUNS the unmodified
``Steady_Elasticity_solver`` (Tools/Steady_solvers.py:13-22) on beam_coarse through the same import harness as
make_golden.py and stores ``d = K^-1 F`` in ``steady_beam_coarse.npz``.  Build container only.

    python tests/golden/make_golden_steady.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (harness: stub meshio / h5py / mpi4py, reference on sys.path)


def main():
    mg.install_harness()
    import Tools.commons as CM
    import Tools.Steady_solvers as SS

    g = np.load(os.path.join(HERE, "beam_coarse_mesh.npz"))
    s = np.load(os.path.join(HERE, "serial_setup.npz"))
    P, C = g["points"], g["tetra"]
    E, nu, rho, fz = 1e6, 0.3, 1, 0.5
    elas_steady = CM.elasticity(E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu)), rho, fz, False)  # Data_prepare.py:161
    dirichlet = CM.node_to_dof(3, [0, 1, 2], list(s["dirichlet_nodes"]))                                 # :141-142
    d = SS.Steady_Elasticity_solver(1, C, P, dirichlet, elas_steady, t=None, Facets=None, Neumann=None)  # :163
    np.savez_compressed(os.path.join(HERE, "steady_beam_coarse.npz"), d_steady=np.asarray(d).reshape(-1),
                        dirichlet_dofs=np.array(sorted(dirichlet)))
    print("steady_beam_coarse.npz: max|d| =", np.abs(d).max(), "tip uz =", d.reshape(-1, 3)[np.argmax(P[:, 0])])


if __name__ == "__main__":
    main()
