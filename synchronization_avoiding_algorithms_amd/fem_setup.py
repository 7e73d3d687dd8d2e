"""Host-side set-up of the explicit solver in O(N): what ``Data_prepare.py:104-204`` does with dense
``(3N)^2`` matrices and O(N^2) list scans, restated with closed forms and vectorised NumPy.

* lumped mass / pre-assembled load: the reference row-sums the consistent mass of
  ``Global_Assembly_no_bc`` (``Tools/Mat_construction.py:199-231``, ``Tools/commons.py:103-107``);
  for linear tets that is ``rho*V_e/4`` per node and ``(V_e/4)*(0,-fz,-fz)``.
* CFL step: ``gamma * 2*min_edge/sqrt(24) / sqrt(E/rho/(1-nu^2))`` (``commons.py:79-90``,
  ``Data_prepare.py:147``).
* partition bookkeeping with the reference's orderings (``Tools/Distributed_tools.py:14-73``):
  local nodes in first-touch order, shared nodes in "other ranks' first-touch" order, sorted
  ``Global_shared``.

The O(N) field set-up also exists as HIP kernels (``device_setup_fields`` -> ``saa_setup_fields``,
``csrc/saa_setup.hip``): that is what :class:`distributed.PartitionedSolver` and ``bench.py`` run, each rank on the
elements touching its own nodes only; the NumPy closed forms below back the drop-in ``Tools`` functions that hand
host arrays around like the reference does.  Nothing in this module imports ``oracle``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def lame(E: float, nu: float):
    """(lambda, mu) as formed at ``Data_prepare.py:47``."""
    return E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))


def signed_volumes(points: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """``detJ/6`` per element, sign kept like the reference (``Mat_construction.py:93``)."""
    p = points[cells]
    e1, e2, e3 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]
    return np.einsum("ij,ij->i", e1, np.cross(e2, e3)) / 6.0


def element_stiffness(points, cells, lmd, mu) -> np.ndarray:
    """``(ne,12,12)`` element stiffness of linear tets in closed form, ``K_e = (detJ/6) B^T D B`` with
    ``grad N_a = c_a/detJ`` (``c_a`` = cofactor rows of the edge matrix) - the matrix the 4-point quadrature of
    ``Local_K_coronary`` (``Mat_construction.py:79-119``) sums to; dof order ``3a + A``, Voigt rows xx,yy,zz,yz,xz,xy
    (``:99-104``).  Host-side set-up only (``Global_Assembly``); the step kernels never form it."""
    p = np.asarray(points, dtype=np.float64)[np.asarray(cells)]
    e1, e2, e3 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]
    c = np.stack([np.cross(e2, e3), np.cross(e3, e1), np.cross(e1, e2)], axis=1)          # (ne,3,3) = detJ * grad N_1..3
    det = np.einsum("ij,ij->i", e1, c[:, 0])
    grad = np.concatenate([-c.sum(axis=1, keepdims=True), c], axis=1) / det[:, None, None]  # (ne,4,3)
    ne = len(grad)
    B = np.zeros((ne, 6, 12))
    gx, gy, gz = grad[..., 0], grad[..., 1], grad[..., 2]
    ix, iy, iz = np.arange(0, 12, 3), np.arange(1, 12, 3), np.arange(2, 12, 3)
    B[:, 0, ix], B[:, 1, iy], B[:, 2, iz] = gx, gy, gz
    B[:, 3, iy], B[:, 3, iz] = gz, gy
    B[:, 4, ix], B[:, 4, iz] = gz, gx
    B[:, 5, ix], B[:, 5, iy] = gy, gx
    D = np.zeros((6, 6))
    D[:3, :3] = lmd
    D[np.arange(3), np.arange(3)] += 2.0 * mu
    D[np.arange(3, 6), np.arange(3, 6)] = mu
    return np.matmul(np.matmul(B.transpose(0, 2, 1), D), B) * (det / 6.0)[:, None, None]


def lumped_mass_and_load(points, cells, rho, fz):
    """(lumped_M, F_pre), each ``(3N,1)`` like the arrays broadcast at ``Data_prepare.py:194-197``."""
    n = len(points)
    quarter = np.repeat(signed_volumes(points, cells) / 4.0, 4)
    nodal = np.bincount(np.asarray(cells).ravel(), weights=quarter, minlength=n)
    lumped = np.repeat(rho * nodal, 3).reshape(-1, 1)
    load = np.outer(nodal, np.array([0.0, -fz, -fz])).reshape(-1, 1)
    return lumped, load


def meshsize(points, cells) -> float:
    p = points[np.asarray(cells)]
    best = np.inf
    for a, b in ((0, 1), (1, 2), (2, 3), (1, 3), (0, 3), (0, 2)):
        d = p[:, a] - p[:, b]
        best = min(best, float(np.sqrt(np.einsum("ij,ij->i", d, d).min())))
    return 2.0 * best / np.sqrt(24)


def cfl_dt(points, cells, E, nu, rho, gamma) -> float:
    return gamma * meshsize(points, cells) / np.sqrt(E / rho / (1 - nu ** 2))


def node_to_dof(nodes) -> np.ndarray:
    """dof = 3*node + component (``commons.py:66-71``)."""
    nodes = np.asarray(nodes, dtype=np.int64)
    return (3 * nodes[:, None] + np.arange(3)[None, :]).ravel()


def first_touch_nodes(cells_of_rank: np.ndarray) -> np.ndarray:
    """Nodes in the order a sweep over the rank's elements first meets them
    (``rankwise_dist``, ``Distributed_tools.py:14-24``)."""
    flat = np.asarray(cells_of_rank).ravel()
    _, first = np.unique(flat, return_index=True)
    return flat[np.sort(first)]


@dataclass
class RankLayout:
    """What one rank knows after ``Data_prepare.py:104-144``."""
    rank: int
    elements: np.ndarray            # global element ids, mesh order      (Local_ele_list)
    nodes: np.ndarray               # global node ids, first-touch order  (Local_nodal_list)
    cells_local: np.ndarray         # (ne,4) int32 local node ids
    shared_nodes: np.ndarray        # global ids, reference order         (shared_nodes)
    shared_local: np.ndarray        # local ids of shared_nodes           (local_mat_node)
    shared_slots: np.ndarray        # positions in Global_shared
    dirichlet_dofs: np.ndarray      # local dofs                          (Local_Dirichlet)
    loc_dof_shared: np.ndarray = field(default=None)  # Online_predictor.py:129

    @property
    def local_dof(self):
        return node_to_dof(self.nodes)


def build_layouts(cells: np.ndarray, epart: np.ndarray, n_parts: int, n_nodes: int,
                  dirichlet_nodes: np.ndarray):
    """All ranks' layouts + ``Global_shared`` (every process can do this: the mesh is replicated,
    as in the reference, ``Data_prepare.py:76-79``)."""
    cells = np.asarray(cells, dtype=np.int64)
    epart = np.asarray(epart)
    elements = [np.nonzero(epart == r)[0] for r in range(n_parts)]
    nodes = [first_touch_nodes(cells[e]) for e in elements]
    member = np.zeros((n_parts, n_nodes), dtype=bool)
    for r in range(n_parts):
        member[r, nodes[r]] = True
    multiplicity = member.sum(axis=0)
    global_shared = np.nonzero(multiplicity > 1)[0]           # sorted union (sort_shared, :44-51)
    slot_of = np.full(n_nodes, -1, dtype=np.int64)
    slot_of[global_shared] = np.arange(len(global_shared))
    is_dirichlet = np.zeros(n_nodes, dtype=bool)
    is_dirichlet[np.asarray(dirichlet_nodes, dtype=np.int64)] = True

    layouts = []
    for r in range(n_parts):
        local_of = np.full(n_nodes, -1, dtype=np.int64)
        local_of[nodes[r]] = np.arange(len(nodes[r]))
        # find_shared_nodes (:29-40): sweep the other ranks' lists in rank order, keep first hits
        hits = [nodes[q][member[r, nodes[q]]] for q in range(n_parts) if q != r]
        if hits and sum(len(h) for h in hits):
            cat = np.concatenate(hits)
            _, first = np.unique(cat, return_index=True)
            shared = cat[np.sort(first)]
        else:
            shared = np.zeros(0, dtype=np.int64)
        shared_local = local_of[shared]
        dloc = np.nonzero(is_dirichlet[nodes[r]])[0]             # Dirichlet_rank_dist (:55-62)
        layouts.append(RankLayout(
            rank=r, elements=elements[r], nodes=nodes[r],
            cells_local=local_of[cells[elements[r]]].astype(np.int32),
            shared_nodes=shared, shared_local=shared_local.astype(np.int32),
            shared_slots=slot_of[shared].astype(np.int32),
            dirichlet_dofs=node_to_dof(dloc).astype(np.int32),
            loc_dof_shared=node_to_dof(shared_local)))
    return layouts, global_shared


def device_setup_fields(points, cells, rho, fz, device=0):
    """Lumped mass ``(3n,1)``, pre-assembled load ``(3n,1)`` and shortest edge of the given elements, computed by the
    HIP set-up kernels (``saa_setup_fields``; replaces ``Global_Assembly_no_bc`` + ``lumping_to_vec`` + ``Meshsize``,
    ``Data_prepare.py:147,175-176``).  ``cells`` index into ``points``."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    pts = np.ascontiguousarray(points, dtype=np.float64)
    tets = np.ascontiguousarray(cells, dtype=np.int32)
    n = len(pts)
    lumped, load, edge = np.empty(3 * n), np.empty(3 * n), C.c_double()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    _lib.check(lib.saa_setup_fields(int(device), n, len(tets), pts.ctypes.data_as(dp),
                                    tets.ctypes.data_as(ip) if len(tets) else None, float(rho), float(fz),
                                    lumped.ctypes.data_as(dp), load.ctypes.data_as(dp), C.byref(edge)))
    return lumped.reshape(-1, 1), load.reshape(-1, 1), edge.value


def host_setup_fields(points, cells, rho, fz, device=0):
    """Same triple from the NumPy closed forms (CPU-only tests of the orchestration inject this)."""
    lumped, load = lumped_mass_and_load(np.asarray(points, dtype=np.float64), np.asarray(cells), rho, fz)
    return lumped, load, meshsize(np.asarray(points, dtype=np.float64), cells) * np.sqrt(24) / 2.0


def dt_from_min_edge(min_edge, E, nu, rho, gamma) -> float:
    """``gamma * Meshsize / sqrt(E/rho/(1-nu^2))`` with ``Meshsize = 2*min_edge/sqrt(24)`` (``commons.py:79-90``,
    ``Data_prepare.py:147``), evaluated in the order :func:`cfl_dt` uses."""
    return gamma * (2.0 * min_edge / np.sqrt(24)) / np.sqrt(E / rho / (1 - nu ** 2))


def build_rank_layout(cells: np.ndarray, epart: np.ndarray, rank: int, n_parts: int, n_nodes: int,
                      dirichlet_nodes: np.ndarray):
    """THIS rank's :class:`RankLayout` and ``Global_shared`` without building anybody else's: O(Ne) passes over the
    replicated mesh (``Data_prepare.py:76-79``) plus sorts over this rank's own elements and the other ranks' elements
    that touch its nodes.  Same orderings as :func:`build_layouts` (the reference's, ``Distributed_tools.py:14-62``)."""
    cells = np.asarray(cells, dtype=np.int64)
    epart = np.asarray(epart)
    elements = np.nonzero(epart == rank)[0]
    nodes = first_touch_nodes(cells[elements])
    mine = np.zeros(n_nodes, dtype=bool)
    mine[nodes] = True
    # how many parts touch a node: one boolean sweep per part (P is the number of GPUs, small)
    count = np.zeros(n_nodes, dtype=np.int16)
    for q in range(n_parts):
        seen = np.zeros(n_nodes, dtype=bool)
        seen[cells[epart == q].ravel()] = True
        count += seen
    global_shared = np.nonzero(count > 1)[0]                      # sorted union (sort_shared, :44-51)
    slot_of = np.full(n_nodes, -1, dtype=np.int64)
    slot_of[global_shared] = np.arange(len(global_shared))
    # find_shared_nodes (:29-40): sweep the other ranks' first-touch lists in rank order, keep first hits.  A node's
    # first occurrence in rank q's sweep lies in an element containing it, so the elements of q that touch one of MY
    # shared nodes carry the whole order.
    cand = mine & (count > 1)
    hits = []
    for q in range(n_parts):
        if q == rank:
            continue
        eq = np.nonzero(epart == q)[0]
        flat = cells[eq[cand[cells[eq]].any(axis=1)]].ravel()
        flat = flat[cand[flat]]
        if flat.size:
            _, first = np.unique(flat, return_index=True)
            hits.append(flat[np.sort(first)])
    if hits:
        cat = np.concatenate(hits)
        _, first = np.unique(cat, return_index=True)
        shared = cat[np.sort(first)]
    else:
        shared = np.zeros(0, dtype=np.int64)
    local_of = np.full(n_nodes, -1, dtype=np.int64)
    local_of[nodes] = np.arange(len(nodes))
    is_dirichlet = np.zeros(n_nodes, dtype=bool)
    is_dirichlet[np.asarray(dirichlet_nodes, dtype=np.int64)] = True
    shared_local = local_of[shared]
    dloc = np.nonzero(is_dirichlet[nodes])[0]
    layout = RankLayout(rank=rank, elements=elements, nodes=nodes, cells_local=local_of[cells[elements]].astype(np.int32),
                        shared_nodes=shared, shared_local=shared_local.astype(np.int32),
                        shared_slots=slot_of[shared].astype(np.int32), dirichlet_dofs=node_to_dof(dloc).astype(np.int32),
                        loc_dof_shared=node_to_dof(shared_local))
    return layout, global_shared


def device_rank_layout(cells, epart, rank, n_parts, n_nodes, dirichlet_nodes=None, points=None, facets=None, device=0,
                       clamp_tol=1e-9):
    """:func:`build_rank_layout` on the GPU (``saa_topology_build``, ``csrc/saa_topology.hip``): holder masks and first-touch
    keys by integer atomics, the reference's orderings by radix sorts.  Dirichlet nodes either as a list
    (``dirichlet_nodes``) or detected on the device from ``points`` and the boundary ``facets`` (triangles with all
    ``|x| < clamp_tol``, ``Data_prepare.py:127-136``).  Returns ``(RankLayout, Global_shared, dirichlet_nodes)``."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    tets = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 4)
    parts = np.ascontiguousarray(epart, dtype=np.int32)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    detect = facets is not None and dirichlet_nodes is None
    if detect:
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        fac = np.ascontiguousarray(facets, dtype=np.int32).reshape(-1, 3)
    h = C.c_void_p()
    _lib.check(lib.saa_topology_build(int(device), int(n_nodes), len(tets), tets.ctypes.data_as(ip) if len(tets) else None,
                                      parts.ctypes.data_as(ip) if len(parts) else None, int(rank), int(n_parts),
                                      pts.ctypes.data_as(dp) if detect and len(fac) else None, len(fac) if detect else 0,
                                      fac.ctypes.data_as(ip) if detect and len(fac) else None, float(clamp_tol), C.byref(h)))
    try:
        sizes = (C.c_int32 * 6)()
        _lib.check(lib.saa_topology_sizes(h, sizes))
        n_le, n_ln, n_sh, n_gs, n_d, n_dl = (int(v) for v in sizes)
        out = [np.empty(n, dtype=np.int32) for n in (n_le, n_ln, 4 * n_le, n_sh, n_sh, n_sh, n_gs, n_d, n_dl)]
        _lib.check(lib.saa_topology_get(h, *[a.ctypes.data_as(ip) if a.size else None for a in out]))
    finally:
        lib.saa_topology_destroy(h)
    elements, nodes, cells_local, shared, shared_local, shared_slots, gshared, dnodes, dlocal = out
    nodes = nodes.astype(np.int64)
    if not detect:
        dnodes = np.asarray(dirichlet_nodes if dirichlet_nodes is not None else [], dtype=np.int64)
        is_d = np.zeros(n_nodes, dtype=bool)
        is_d[dnodes] = True
        dlocal = np.nonzero(is_d[nodes])[0]
    layout = RankLayout(rank=rank, elements=elements.astype(np.int64), nodes=nodes, cells_local=cells_local.reshape(-1, 4),
                        shared_nodes=shared.astype(np.int64), shared_local=shared_local, shared_slots=shared_slots,
                        dirichlet_dofs=node_to_dof(dlocal).astype(np.int32), loc_dof_shared=node_to_dof(shared_local))
    return layout, gshared.astype(np.int64), np.asarray(dnodes, dtype=np.int64)


def rank_fields(points, cells, layout: RankLayout, rho, fz, device=0, setup=None):
    """``l_M``, ``F_rankwise`` ``(3 n_local, 1)`` of this rank (``Data_prepare.py:200-202``: the GLOBAL lumped mass and
    load restricted to the rank's nodes, so shared nodes carry the contributions of the other ranks' elements too) and
    the shortest edge among the elements involved.  Only the elements touching this rank's nodes are looked at."""
    setup = setup or device_setup_fields
    cells = np.asarray(cells, dtype=np.int64)
    mine = np.zeros(len(points), dtype=bool)
    mine[layout.nodes] = True
    touching = cells[mine[cells].any(axis=1)]
    sub_nodes = np.unique(touching)
    sub_of = np.full(len(points), -1, dtype=np.int64)
    sub_of[sub_nodes] = np.arange(len(sub_nodes))
    lumped, load, min_edge = setup(np.asarray(points, dtype=np.float64)[sub_nodes], sub_of[touching], rho, fz, device)
    dof = node_to_dof(sub_of[layout.nodes])
    return lumped[dof], load[dof], min_edge


def rank_problem(points, cells, dirichlet_nodes, epart, rank, n_parts, E, nu, rho, fz, gamma, device=0, setup=None, facets=None):
    """Everything rank ``rank`` needs to create its solver (``Data_prepare.py:104-204`` for one rank): its layout,
    ``Global_shared``, ``l_M`` / ``F_rankwise`` from the set-up kernels and its LOCAL CFL step - the caller takes the
    minimum over the ranks (``Data_prepare.py:148-154``).  On the GPU (``setup`` left at its default) the layout comes from
    the topology kernels (:func:`device_rank_layout`); with ``dirichlet_nodes=None`` and the boundary ``facets`` given the
    clamped nodes are detected there too.  A host ``setup`` (CPU tests of the orchestration) keeps the NumPy layout."""
    if setup is None:
        layout, global_shared, _ = device_rank_layout(cells, epart, rank, n_parts, len(points), dirichlet_nodes, points, facets,
                                                      device)
    else:
        if dirichlet_nodes is None:
            raise ValueError("the host layout needs the list of clamped nodes")
        layout, global_shared = build_rank_layout(cells, epart, rank, n_parts, len(points), dirichlet_nodes)
    l_M, F_rankwise, min_edge = rank_fields(points, cells, layout, rho, fz, device, setup)
    return layout, global_shared, l_M, F_rankwise, dt_from_min_edge(min_edge, E, nu, rho, gamma)
