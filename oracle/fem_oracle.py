"""ORACLE (test infrastructure, NOT product code) - CPU restatement of the reference hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package must never do so.

What it restates (all citations are into /root/reference):

* element operators      ``Tools/Mat_construction.py:23-119`` with ``Tools/Shape_function_Deriv.py:9-12,
                         33-36,60-67`` and ``Tools/Qudrature.py:7-12``
* per-rank stiffness     ``Tools/Mat_construction.py:122-150`` (as scipy CSR built from COO triplets
                         instead of a dense (3n)^2 array - same entries, summed in element order)
* lumped mass / F_pre    ``Tools/Mat_construction.py:199-231`` + ``Tools/commons.py:103-107``
* CFL step               ``Tools/commons.py:79-90`` + ``Data_prepare.py:147``
* partition helpers      ``Tools/Distributed_tools.py:14-73``
* one explicit step      ``Tools/Dynamic_solver.py:9-34`` (identical expression, identical association)
* shared-node sum        ``Tools/Distributed_tools.py:77-92``
* time loops             ``Data_prepare.py:223-240`` and ``Online_predictor.py:251-318``

Pinning: every function here is checked in ``tests/test_oracle_golden.py`` against vectors written by
``tests/golden/make_golden.py``, which imports and runs the unmodified reference modules in the build
container (see DESIGN.md "Oracle").  The LSTM part lives in ``oracle/lstm_oracle.py``.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix

# ----------------------------------------------------------------------------------------------
# Element level
# ----------------------------------------------------------------------------------------------

#: 4-point rule used for p=1 (Qudrature.py:7-12): nodes and weights 0.25/6
_A, _B = 0.5854101966249685, 0.1381966011250105
QUAD_NODES = np.array([[_A, _B, _B], [_B, _A, _B], [_B, _B, _A], [_B, _B, _B]])
QUAD_WEIGHTS = np.array([0.25 / 6] * 4)

#: dN/dxi for the linear tet (Shape_function_Deriv.py:36)
DN_DXI = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])


def shape_functions(xi):
    """Shape_function_Deriv.py:9-12 (p = 1)."""
    x, y, z = xi
    return np.array([1.0 - x - y - z, x, y, z])


def lame(E, nu):
    """Data_prepare.py:47 - (lambda, mu) handed to ``elasticity``."""
    return E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))


def elasticity_D(lmd, mu):
    """commons.py:25-31 - isotropic 6x6 matrix in (xx,yy,zz,yz,xz,xy) order."""
    D = np.zeros((6, 6))
    D[:3, :3] = lmd
    D[np.arange(3), np.arange(3)] = lmd + 2.0 * mu
    D[np.arange(3, 6), np.arange(3, 6)] = mu
    return D


def jacobians(P):
    """Shape_function_Deriv.py:60-67 for a batch: ``J[e,i,j] = sum_a dN_a/dxi_j * P[e,a,i]``."""
    return np.einsum("aj,eai->eij", DN_DXI, P)


def physical_gradients(P):
    """``N_xyz = Shape_Deri @ inv(J)`` (Mat_construction.py:94-96) and signed detJ (:93)."""
    J = jacobians(P)
    detJ = np.linalg.det(J)
    invJ = np.linalg.inv(J)
    return np.einsum("aj,ejk->eak", DN_DXI, invJ), detJ


def b_matrices(grad):
    """B_a rows (xx,yy,zz,yz,xz,xy) - Mat_construction.py:99-104.  grad: (ne,4,3) -> (ne,4,6,3)."""
    ne = grad.shape[0]
    B = np.zeros((ne, 4, 6, 3))
    gx, gy, gz = grad[..., 0], grad[..., 1], grad[..., 2]
    B[:, :, 0, 0] = gx
    B[:, :, 1, 1] = gy
    B[:, :, 2, 2] = gz
    B[:, :, 3, 1], B[:, :, 3, 2] = gz, gy
    B[:, :, 4, 0], B[:, :, 4, 2] = gz, gx
    B[:, :, 5, 0], B[:, :, 5, 1] = gy, gx
    return B


def element_stiffness(P, lmd, mu):
    """``Local_K_coronary`` (Mat_construction.py:79-119) for a batch of elements.

    Evaluated as the literal 4-point quadrature sum ``sum_q Bi^T D Bj detJ w_q`` (B and detJ are
    constant for p=1, so the four terms are equal; they are still added one by one like the
    reference does).  P: (ne,4,3) -> (ne,12,12), dof order 3*a + A.
    """
    P = np.asarray(P, dtype=np.float64).reshape(-1, 4, 3)
    grad, detJ = physical_gradients(P)
    B = b_matrices(grad)
    D = elasticity_D(lmd, mu)
    # all sixteen blocks (Bi^T D) Bj at once: B as (ne,6,12) with column 3a+i, same association as :112
    Bf = np.ascontiguousarray(B.transpose(0, 2, 1, 3)).reshape(-1, 6, 12)
    BtDB = np.matmul(np.matmul(Bf.transpose(0, 2, 1), D), Bf)  # (ne,12,12)
    K = np.zeros_like(BtDB)
    for w in QUAD_WEIGHTS:
        K += BtDB * detJ[:, None, None] * w
    return K


def element_mass_force(P, rho, fz):
    """``Local_MKF`` mass and load parts (Mat_construction.py:23-76), un-ramped load (R=False).

    Consistent mass ``M[3i+A,3j+A] = sum_q N_i rho N_j detJ w`` and ``F[3i+C] = sum_q N_i f_C detJ w``
    with ``f = (0,-fz,-fz)`` (commons.py:35-41).
    """
    P = np.asarray(P, dtype=np.float64).reshape(-1, 4, 3)
    _, detJ = physical_gradients(P)
    ne = P.shape[0]
    M = np.zeros((ne, 12, 12))
    F = np.zeros((ne, 12))
    f_loc = np.array([0.0, -fz, -fz])
    for q in range(4):
        N = shape_functions(QUAD_NODES[q])
        for i in range(4):
            for j in range(4):
                m = N[i] * rho * N[j] * detJ * QUAD_WEIGHTS[q]
                for A in range(3):
                    M[:, 3 * i + A, 3 * j + A] += m
            for C in range(3):
                F[:, 3 * i + C] += N[i] * f_loc[C] * detJ * QUAD_WEIGHTS[q]
    return M, F


# ----------------------------------------------------------------------------------------------
# Assembly
# ----------------------------------------------------------------------------------------------

def node_to_dof(nodes):
    """commons.py:66-71 with d=3, ls=[0,1,2]: dof = 3*node + comp (node-interleaved)."""
    nodes = np.asarray(nodes, dtype=np.int64)
    return (3 * nodes[:, None] + np.arange(3)[None, :]).ravel()


def local_index(global_ids, local_node_list):
    """``local_mat_node`` (Distributed_tools.py:66-73): position of each id in the local list."""
    lut = {int(g): i for i, g in enumerate(local_node_list)}
    return np.array([lut[int(g)] for g in np.asarray(global_ids).ravel()], dtype=np.int64).reshape(
        np.shape(global_ids))


def assemble_local_stiffness(local_node_list, cells, points, lmd, mu):
    """``Local_assembly_for_stiffness`` (Mat_construction.py:122-150) -> scipy CSR, local numbering.

    ``cells`` hold *global* node ids of this rank's elements; rows/cols follow ``local_node_list``
    (first-touch order).  No boundary-condition elimination, exactly like the reference.
    """
    cells = np.asarray(cells, dtype=np.int64)
    n = len(local_node_list)
    Ke = element_stiffness(points[cells], lmd, mu)
    loc = local_index(cells, local_node_list)  # (ne,4)
    dof = (3 * loc[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 12)
    rows = np.repeat(dof, 12, axis=1).ravel()
    cols = np.tile(dof, (1, 12)).ravel()
    # accumulate duplicates strictly in element order, like the dense ``K[P,Q] +=`` of :148
    key = rows * (3 * n) + cols
    ukey, inv = np.unique(key, return_inverse=True)
    vals = np.zeros(len(ukey))
    np.add.at(vals, inv, Ke.ravel())
    keep = vals != 0.0  # csr_matrix(dense) drops exact zeros (:150)
    ukey, vals = ukey[keep], vals[keep]
    K = csr_matrix((vals, (ukey // (3 * n), ukey % (3 * n))), shape=(3 * n, 3 * n))
    K.sort_indices()
    return K


def assemble_local_stiffness_blocked(local_node_list, cells, points, lmd, mu, chunk=65536):
    """``Local_assembly_for_stiffness`` (Mat_construction.py:122-150) for meshes where :func:`assemble_local_stiffness`'
    sort of all 144 scalar entries per element is too slow (1M tets: 148 M COO triplets): the sparsity pattern is found
    on NODE pairs (16 per element), the 3x3 blocks of the element matrices are accumulated into it (``np.add.at``) strictly in
    element order - the order of the reference's dense ``K[P,Q] += ...`` (:148) - and the result is the same CSR matrix, bit for
    bit (tests/test_oracle_golden.py compares the two), with explicit zeros dropped like ``csr_matrix(dense)`` does
    (:150).  ``cells``: global node ids of this rank's elements; rows/cols follow ``local_node_list``."""
    from scipy.sparse import bsr_matrix

    cells = np.asarray(cells, dtype=np.int64)
    n = len(local_node_list)
    lut = np.full(int(max(np.max(local_node_list), cells.max())) + 1, -1, dtype=np.int64)
    lut[np.asarray(local_node_list, dtype=np.int64)] = np.arange(n)
    loc = lut[cells]                                                     # (ne,4) local node ids
    pair = (loc[:, :, None] * n + loc[:, None, :]).reshape(-1)           # (ne*16) key of node pair (a, b), row-major
    upair, inv = np.unique(pair, return_inverse=True)
    blocks = np.zeros((len(upair), 3, 3))
    for c0 in range(0, len(cells), chunk):  # chunks in element order; ufunc.at adds in index order, i.e. element order
        Ke = element_stiffness(points[cells[c0:c0 + chunk]], lmd, mu).reshape(-1, 4, 3, 4, 3)
        Kb = np.ascontiguousarray(Ke.transpose(0, 1, 3, 2, 4)).reshape(-1, 3, 3)   # block (a, b) of element e at 16e+4a+b
        np.add.at(blocks, inv[16 * c0:16 * (c0 + len(Ke))], Kb)
    indptr = np.searchsorted(upair // n, np.arange(n + 1))
    K = bsr_matrix((blocks, upair % n, indptr), shape=(3 * n, 3 * n)).tocsr()
    K.eliminate_zeros()
    K.sort_indices()
    return K


class MatrixFreeStiffness:
    """``LocalK`` for meshes whose assembled matrix is too expensive to build on the test host (1M tets: 148 M COO
    triplets): the same element matrices ``Local_K_coronary`` produces (Mat_construction.py:79-119, via
    :func:`element_stiffness`), kept per element and applied as ``sum_e scatter(K_e . gather(d))`` - what
    ``Local_assembly_for_stiffness`` (:122-150) followed by ``LocalK.dot`` (Dynamic_solver.py:12) computes, with the
    additions re-associated (per element first, then over the elements around a dof in element order).  Pinned to
    the reference's own ``LocalK.dot(d)`` on beam_coarse in ``tests/test_oracle_golden.py``.

    ``cells_local``: (ne,4) node ids in the numbering of ``points_local``; 1.15 kB per element.
    """

    def __init__(self, cells_local, points_local, lmd, mu, chunk=65536, threads=None):
        import os
        from concurrent.futures import ThreadPoolExecutor

        cells = np.asarray(cells_local, dtype=np.int64)
        pts = np.asarray(points_local, dtype=np.float64)
        self.n_dof = 3 * len(pts)
        self.dof = (3 * cells[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 12)
        self.Ke = np.empty((len(cells), 12, 12))
        self.shape = (self.n_dof, self.n_dof)
        # element ranges are independent; NumPy releases the GIL inside its loops.  The partial vectors of the
        # ranges are added in range order, so the result does not depend on thread timing.
        self._pool = ThreadPoolExecutor(max(1, min(threads or 8, os.cpu_count() or 1)))
        self._ranges = [(lo, min(lo + chunk, len(cells))) for lo in range(0, len(cells), chunk)]

        def fill(r):
            self.Ke[r[0]:r[1]] = element_stiffness(pts[cells[r[0]:r[1]]], lmd, mu)

        list(self._pool.map(fill, self._ranges))

    def dot(self, d):
        d = np.asarray(d, dtype=np.float64).reshape(-1)

        def part(r):
            dof = self.dof[r[0]:r[1]]
            fe = np.matmul(self.Ke[r[0]:r[1]], d[dof][:, :, None])[:, :, 0]
            return np.bincount(dof.ravel(), weights=fe.ravel(), minlength=self.n_dof)

        out = np.zeros(self.n_dof)
        for p in self._pool.map(part, self._ranges):
            out += p
        return out.reshape(-1, 1)


def lumped_mass_and_load(cells, points, rho, fz):
    """``Global_Assembly_no_bc`` + ``lumping_to_vec`` (Mat_construction.py:199-231, commons.py:103-107).

    Returns (lumped_M, F_pre), both (3N,1): row sums of the consistent mass and the pre-assembled
    un-ramped body force, accumulated element by element in mesh order.
    """
    cells = np.asarray(cells, dtype=np.int64)
    N = len(points)
    Me, Fe = element_mass_force(points[cells], rho, fz)
    dof = (3 * cells[:, :, None] + np.arange(3)[None, None, :]).reshape(-1, 12)
    lumped = np.zeros(3 * N)
    np.add.at(lumped, dof.ravel(), Me.sum(axis=2).ravel())
    F = np.zeros(3 * N)
    np.add.at(F, dof.ravel(), Fe.ravel())
    return lumped.reshape(-1, 1), F.reshape(-1, 1)


def steady_solve(cells, points, dirichlet_dofs, lmd, mu, rho, fz):
    """``Steady_Elasticity_solver`` (Steady_solvers.py:13-22) with ``Global_Assembly`` (Mat_construction.py:154-196):
    rows and columns of Dirichlet dofs are never assembled, their diagonal is set to 1 and their load to 0, then
    ``d = K^-1 F`` with the un-ramped load.  Sparse direct solve here instead of the dense one."""
    from scipy.sparse import identity
    from scipy.sparse.linalg import spsolve

    n = len(points)
    K = assemble_local_stiffness(np.arange(n), cells, points, lmd, mu)
    _, F = lumped_mass_and_load(cells, points, rho, fz)
    free = np.ones(3 * n)
    free[np.asarray(dirichlet_dofs, dtype=np.int64)] = 0.0
    from scipy.sparse import diags

    Dm = diags(free)
    A = (Dm @ K @ Dm + diags(1.0 - free)).tocsc()
    return spsolve(A, F.ravel() * free).reshape(-1, 1)


def meshsize(cells, points):
    """``Meshsize`` (commons.py:79-90): 2*min_edge/sqrt(24) over the given elements."""
    P = points[np.asarray(cells, dtype=np.int64)]
    pairs = [(0, 1), (1, 2), (2, 3), (1, 3), (0, 3), (0, 2)]
    lens = np.stack([np.linalg.norm(P[:, a] - P[:, b], axis=1) for a, b in pairs], axis=1)
    return 2.0 * lens.min() / np.sqrt(24)


def cfl_dt(cells, points, E, nu, rho, gamma):
    """Data_prepare.py:147."""
    return gamma * meshsize(cells, points) / np.sqrt(E / rho / (1 - nu ** 2))


# ----------------------------------------------------------------------------------------------
# Partition bookkeeping (Distributed_tools.py:14-62)
# ----------------------------------------------------------------------------------------------

def rankwise_dist(rank, epart, cells):
    """Elements of ``rank`` in mesh order and their nodes in first-touch order (:14-24)."""
    ele = np.nonzero(np.asarray(epart) == rank)[0]
    flat = np.asarray(cells)[ele].ravel()
    _, first = np.unique(flat, return_index=True)
    return ele, flat[np.sort(first)]


def find_shared_nodes(rank, rank_nodal_list):
    """Nodes of ``rank`` that some other rank also holds, ordered by the other ranks' lists (:29-40)."""
    mine = set(int(v) for v in rank_nodal_list[rank])
    out, seen = [], set()
    for r, lst in enumerate(rank_nodal_list):
        if r == rank:
            continue
        for idx in lst:
            idx = int(idx)
            if idx in mine and idx not in seen:
                seen.add(idx)
                out.append(idx)
    return np.array(out, dtype=np.int64)


def sort_shared(per_rank_shared):
    """Sorted union (:44-51)."""
    if len(per_rank_shared) == 0:
        return np.zeros(0, dtype=np.int64)
    return np.unique(np.concatenate([np.asarray(s, dtype=np.int64) for s in per_rank_shared]))


def dirichlet_rank_dist(dirichlet_nodes, local_node_list):
    """Local dofs of clamped nodes, in local-node order (:55-62)."""
    dn = set(int(v) for v in dirichlet_nodes)
    loc = [i for i, g in enumerate(local_node_list) if int(g) in dn]
    return node_to_dof(loc) if loc else np.zeros(0, dtype=np.int64)


def clamp_nodes(points, facets, tol=1e-9):
    """Data_prepare.py:127-136 - nodes of facets with all |x| < tol, first-seen order."""
    out, seen = [], set()
    for f in np.asarray(facets):
        if all(abs(points[k][0]) < tol for k in f):
            for k in f:
                if int(k) not in seen:
                    seen.add(int(k))
                    out.append(int(k))
    return np.array(out, dtype=np.int64)


def syn_sum(rank_forces, rank_node_lists, n_global):
    """``syn_cpus`` (Distributed_tools.py:77-92) without MPI: add in rank order, then restrict."""
    f_global = np.zeros((3 * n_global, 1))
    for f, nodes in zip(rank_forces, rank_node_lists):
        f_global[node_to_dof(nodes)] += f
    return [f_global[node_to_dof(nodes)] for nodes in rank_node_lists]


# ----------------------------------------------------------------------------------------------
# Time stepping
# ----------------------------------------------------------------------------------------------

def linear_ramp(t):
    """commons.py:7-11."""
    return t if t <= 1 else 1.0


def cd_update(F_int, F_rankwise, l_M, d0, dn, dt, tn, alpha, dirichlet):
    """Dynamic_solver.py:13-20 - the update expression in the reference's association order."""
    F_ext = (F_rankwise * linear_ramp(tn)).reshape((len(F_rankwise), 1))
    l_M = l_M.reshape((len(l_M), 1))
    d1 = (dt ** 2 * (F_ext - F_int) + 2 * l_M * d0 - l_M * dn + dt / 2 * l_M * alpha * dn) / (
        l_M + 0.5 * alpha * l_M * dt)
    d1[dirichlet] = 0
    return d1


def explicit_step(K, F_rankwise, dirichlet, tn, dt, d0, dn, l_M, alpha):
    """Serial / MODEL=True form of ``parallel_explicit_solver_dis_pre`` (Dynamic_solver.py:9-34)."""
    return cd_update(K.dot(d0), F_rankwise, l_M, d0, dn, dt, tn, alpha, dirichlet)


def explicit_step_synced(Ks, F_ranks, dirichlets, tn, dt, d0s, dns, l_Ms, alpha, node_lists, n_global):
    """MODEL=False, size != 1: every rank's ``K d0`` is summed over shared nodes before the update."""
    partial = [K.dot(d0) for K, d0 in zip(Ks, d0s)]
    summed = syn_sum(partial, node_lists, n_global)
    return [cd_update(f, F, m, d0, dn, dt, tn, alpha, dr)
            for f, F, m, d0, dn, dr in zip(summed, F_ranks, l_Ms, d0s, dns, dirichlets)]


class RankProblem:
    """Everything one rank holds after the set-up of Data_prepare.py:104-209."""

    def __init__(self, rank, epart, cells, points, dirichlet_nodes, lumped_M, F_pre, lmd, mu):
        self.rank = rank
        self.ele, self.nodes = rankwise_dist(rank, epart, cells)
        self.local_dof = node_to_dof(self.nodes)
        self.dirichlet = dirichlet_rank_dist(dirichlet_nodes, self.nodes)
        self.F = F_pre[self.local_dof]
        self.l_M = lumped_M[self.local_dof]
        self.cells = np.asarray(cells)[self.ele]
        # (large partitions: the node-pair formulation of the same assembly, bit-identical and 8x faster)
        assemble = assemble_local_stiffness if len(self.cells) <= 20000 else assemble_local_stiffness_blocked
        self.K = assemble(self.nodes, self.cells, points, lmd, mu)
        self.n_global = len(points)


def setup_problem(points, cells, facets, n_parts, epart, E=1e6, nu=0.3, rho=1.0, fz=0.5, gamma=0.9):
    """Data_prepare.py:104-209 re-enacted: returns (rank problems, dt, per-rank shared nodes, Global_shared)."""
    lmd, mu = lame(E, nu)
    dnodes = clamp_nodes(points, facets)
    lumped_M, F_pre = lumped_mass_and_load(cells, points, rho, fz)
    ranks = [RankProblem(r, epart, cells, points, dnodes, lumped_M, F_pre, lmd, mu) for r in range(n_parts)]
    dt = min(cfl_dt(rp.cells, points, E, nu, rho, gamma) for rp in ranks)
    lists = [rp.nodes for rp in ranks]
    shared = [find_shared_nodes(r, lists) for r in range(n_parts)]
    return ranks, dt, shared, sort_shared(shared)


def run_ground_truth(ranks, dt, n_steps, alpha=0.5, snapshots=()):
    """Data_prepare.py:211-240 for all ranks at once (zero initial state, ghost step zero)."""
    d0s = [np.zeros((len(rp.local_dof), 1)) for rp in ranks]
    dns = [np.zeros((len(rp.local_dof), 1)) for rp in ranks]
    tn = 0
    out = {}
    for i in range(n_steps):
        if len(ranks) == 1:
            rp = ranks[0]
            d1s = [explicit_step(rp.K, rp.F, rp.dirichlet, tn, dt, d0s[0], dns[0], rp.l_M, alpha)]
        else:
            d1s = explicit_step_synced([rp.K for rp in ranks], [rp.F for rp in ranks],
                                       [rp.dirichlet for rp in ranks], tn, dt, d0s, dns,
                                       [rp.l_M for rp in ranks], alpha, [rp.nodes for rp in ranks],
                                       ranks[0].n_global)
        dns, d0s = d0s, d1s
        tn = tn + dt
        if (i + 1) in snapshots:
            out[i + 1] = [d.copy() for d in d1s]
    return d0s, dns, tn, out


def run_hybrid(ranks, dt, n_steps, shared_local_dofs, predictor, n_past, n_future, filter_size,
               alpha=0.5, resync_every=None, resync_steps=None):
    """Online_predictor.py:251-318 for all ranks.

    ``predictor(rank, n, d_sol_shared) -> (n_future*filter_size, input_size)`` plays
    ``encoder_decoder_predictor``.  Returns per-rank saved trajectories (n_dof, n_steps) and the
    shared-dof histories (n_steps, input_size).

    ``resync_every`` (None: the reference, which never synchronises again after the warm-up) is the extension
    BASELINE.json's configs[4] names ("RCCL every k-th step only"), not reference behaviour: after every
    ``resync_every`` predicted windows the next ``resync_steps`` steps (default: one window) are synchronised ones,
    recorded in the history like the warm-up's, so that the following windows are predicted from true values again.
    They start from the mean of the holders' copies of every shared node (d^n and d^(n-1)).
    """
    if resync_steps is None:
        resync_steps = n_future * filter_size
    resync_left, windows = 0, 0
    P = len(ranks)
    i_cri = n_past * filter_size - 1
    d0s = [np.zeros((len(rp.local_dof), 1)) for rp in ranks]
    dns = [np.zeros((len(rp.local_dof), 1)) for rp in ranks]
    hist = [np.zeros((n_steps, len(s))) for s in shared_local_dofs]
    save = [np.zeros((len(rp.local_dof), n_steps)) for rp in ranks]
    tn = 0
    i = 0
    counter2 = 0
    while i < n_steps:
        if i <= i_cri or resync_left > 0:
            resync_left = max(resync_left - 1, 0)
            if P == 1:
                rp = ranks[0]
                d1s = [explicit_step(rp.K, rp.F, rp.dirichlet, tn, dt, d0s[0], dns[0], rp.l_M, alpha)]
            else:
                d1s = explicit_step_synced([rp.K for rp in ranks], [rp.F for rp in ranks],
                                           [rp.dirichlet for rp in ranks], tn, dt, d0s, dns,
                                           [rp.l_M for rp in ranks], alpha, [rp.nodes for rp in ranks],
                                           ranks[0].n_global)
            for r in range(P):
                hist[r][i, :] = d1s[r][shared_local_dofs[r], 0]
                save[r][:, i] = d1s[r][:, 0]
            dns, d0s = d0s, d1s
            tn = tn + dt
            i += 1
        else:
            tables = [predictor(r, i, hist[r]) for r in range(P)]
            start = i
            for k in range(start, start + n_future * filter_size):
                if k >= n_steps:
                    break
                d1s = []
                for r, rp in enumerate(ranks):
                    d1 = explicit_step(rp.K, rp.F, rp.dirichlet, tn, dt, d0s[r], dns[r], rp.l_M, alpha)
                    row = k - start  # (= k - i_cri - 1 - n_future * filter_size * counter2 in the reference's counting)
                    d1[shared_local_dofs[r]] = tables[r][row, :].reshape((-1, 1))
                    hist[r][i, :] = d1[shared_local_dofs[r], 0]
                    save[r][:, i] = d1[:, 0]
                    d1s.append(d1)
                dns, d0s = d0s, d1s
                i += 1
                tn = tn + dt
            counter2 += 1
            windows += 1
            if resync_every and windows % resync_every == 0:
                resync_left = resync_steps
                # the holders' copies of a shared node have gone their own ways (every rank wrote ITS model's values): they
                # continue from their mean, d^n and d^(n-1) alike (distributed.PartitionedSolver.reconcile_shared)
                for d in (d0s, dns):
                    tot, cnt = np.zeros(3 * ranks[0].n_global), np.zeros(3 * ranks[0].n_global)
                    gdofs = []
                    for r, rp in enumerate(ranks):
                        ld = np.asarray(shared_local_dofs[r])
                        gd = 3 * np.asarray(rp.nodes)[ld // 3] + ld % 3
                        gdofs.append((ld, gd))
                        np.add.at(tot, gd, d[r][ld, 0])
                        np.add.at(cnt, gd, 1.0)
                    for r, (ld, gd) in enumerate(gdofs):
                        d[r] = d[r].copy()
                        d[r][ld, 0] = tot[gd] / cnt[gd]
    return save, hist
