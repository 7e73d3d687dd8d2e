"""Mesh input for the explicit linear-tet solver: legacy-VTK reader, synthetic beams, partitions.

The reference reads ``Mesh_info/beam_coarse.vtk`` through meshio
(reference ``Data_prepare.py:56-62``) and partitions elements with ParMETIS
(``Data_prepare.py:82-101``).  Neither library is available to this build, so this
module provides

* :func:`read_vtk` – a reader for the legacy ASCII ``UNSTRUCTURED_GRID`` files Gmsh
  writes (only what the path needs: points, ``tetra`` cells and ``triangle`` facets),
* :func:`structured_beam` – the deterministic synthetic ``25n x n x n`` cantilever of
  ``SURVEY.md`` §8(d): cubes split into 6 positively oriented Kuhn tetrahedra over the
  box of ``Mesh_info/beam_US.geo:2-16``,
* :func:`delaunay_beam` – the same box meshed without any lattice (Delaunay tetrahedra of random points),
* :func:`slab_partition` / :func:`rcb_partition` – element partition vectors playing the
  role of ``part_mesh_kway``'s ``epart`` (one part per GPU).

Everything here is vectorised NumPy and O(N log N) at worst.
"""
from __future__ import annotations

import numpy as np

# VTK cell type ids -> meshio-style names (the reference indexes cells_dict by name,
# Data_prepare.py:59-60)
_VTK_TYPES = {1: "vertex", 3: "line", 5: "triangle", 10: "tetra"}


class Mesh:
    """Minimal stand-in for the object ``meshio.read`` returns (``.points``, ``.cells_dict``)."""

    def __init__(self, points, cells_dict):
        self.points = np.ascontiguousarray(points, dtype=np.float64)
        self.cells_dict = {k: np.ascontiguousarray(v, dtype=np.int64) for k, v in cells_dict.items()}

    @property
    def tets(self):
        return self.cells_dict["tetra"]

    @property
    def triangles(self):
        return self.cells_dict.get("triangle", np.zeros((0, 3), dtype=np.int64))


def read_vtk(path) -> Mesh:
    """Parse a legacy ASCII VTK unstructured grid (the format of ``beam_coarse.vtk``)."""
    with open(path, "r") as fh:
        tok = fh.read().split()
    up = [t.upper() for t in tok]
    try:
        ip = up.index("POINTS")
        ic = up.index("CELLS")
        it = up.index("CELL_TYPES")
    except ValueError as exc:  # pragma: no cover - malformed input
        raise ValueError(f"{path}: not a legacy VTK UNSTRUCTURED_GRID file") from exc
    if "ASCII" not in up[:ic]:
        raise ValueError(f"{path}: only ASCII legacy VTK is supported")
    n_pts = int(tok[ip + 1])
    pts = np.array(tok[ip + 3: ip + 3 + 3 * n_pts], dtype=np.float64).reshape(n_pts, 3)
    n_cells, n_ints = int(tok[ic + 1]), int(tok[ic + 2])
    flat = np.array(tok[ic + 3: ic + 3 + n_ints], dtype=np.int64)
    types = np.array(tok[it + 2: it + 2 + n_cells], dtype=np.int64)
    if int(tok[it + 1]) != n_cells:
        raise ValueError(f"{path}: CELL_TYPES count differs from CELLS count")
    # walk the ragged CELLS list once
    sizes = np.empty(n_cells, dtype=np.int64)
    starts = np.empty(n_cells, dtype=np.int64)
    pos = 0
    for c in range(n_cells):
        sizes[c] = flat[pos]
        starts[c] = pos + 1
        pos += 1 + flat[pos]
    cells = {}
    for vtk_id, name in _VTK_TYPES.items():
        sel = np.nonzero(types == vtk_id)[0]
        if sel.size == 0:
            continue
        width = int(sizes[sel[0]])
        idx = starts[sel][:, None] + np.arange(width)[None, :]
        cells[name] = flat[idx]
    return Mesh(pts, cells)


# Kuhn (Freudenthal) split of the unit cube: one tet per axis permutation, each a
# lattice path 000 -> 111.  Odd permutations get two vertices swapped so that
# det[x1-x0, x2-x0, x3-x0] > 0 (the reference keeps detJ signed,
# Mat_construction.py:93, and beam_coarse has detJ > 0 everywhere).
def _kuhn_corners():
    import itertools

    tets = []
    for perm in itertools.permutations(range(3)):
        v = np.zeros(3, dtype=np.int64)
        path = [v.copy()]
        for ax in perm:
            v[ax] += 1
            path.append(v.copy())
        path = np.array(path)
        jac = (path[1:] - path[0]).T.astype(float)
        if np.linalg.det(jac) < 0:
            path[[2, 3]] = path[[3, 2]]
        tets.append(path)
    return np.array(tets)  # (6, 4, 3) corner offsets


def structured_beam(n: int, length: float = 25.0, width: float = 1.0, height: float = 1.0) -> Mesh:
    """``25n x n x n`` cubes x 6 Kuhn tets on ``[0,L]x[0,W]x[0,H]``.

    Node ids are lexicographic ``(ix*(ny+1) + iy)*(nz+1) + iz``; the ``x = 0`` face is
    triangulated too so that the reference's clamp detection (``Data_prepare.py:127-136``)
    has facets to look at.  n=19 -> 1 028 850 tets / 190 400 nodes; n=38 -> 8 230 800 / 1 446 471.
    """
    if n < 1:
        raise ValueError("n must be >= 1")
    nx, ny, nz = int(round(length / width)) * n, n, n
    gx = np.linspace(0.0, length, nx + 1)
    gy = np.linspace(0.0, width, ny + 1)
    gz = np.linspace(0.0, height, nz + 1)
    X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)

    def nid(ix, iy, iz):
        return (ix * (ny + 1) + iy) * (nz + 1) + iz

    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    corners = _kuhn_corners()  # (6,4,3)
    tets = np.empty((ci.size, 6, 4), dtype=np.int64)
    for t in range(6):
        for a in range(4):
            o = corners[t, a]
            tets[:, t, a] = nid(ci + o[0], cj + o[1], ck + o[2])
    tets = tets.reshape(-1, 4)

    # facets on x = 0: two triangles per boundary square
    fj, fk = np.meshgrid(np.arange(ny), np.arange(nz), indexing="ij")
    fj, fk = fj.ravel(), fk.ravel()
    z0 = np.zeros_like(fj)
    tri = np.concatenate([
        np.stack([nid(z0, fj, fk), nid(z0, fj + 1, fk), nid(z0, fj + 1, fk + 1)], axis=1),
        np.stack([nid(z0, fj, fk), nid(z0, fj + 1, fk + 1), nid(z0, fj, fk + 1)], axis=1),
    ])
    return Mesh(pts, {"tetra": tets, "triangle": tri})


def delaunay_beam(n: int, length: float = 25.0, width: float = 1.0, height: float = 1.0, seed: int = 0,
                  sliver: float = 0.02, density: float = 1.0) -> Mesh:
    """An UNSTRUCTURED mesh of the box of ``Mesh_info/beam_US.geo:2-16`` at the scale of :func:`structured_beam` ``(n)``:
    Delaunay tetrahedra (Qhull through SciPy) of random points - no lattice anywhere in the connectivity (node valences
    1...45, ~5 tets per node), the class of mesh Gmsh writes for the reference (``Mesh_info/beam_coarse.vtk``).

    Points: a hard-core random process in the interior (uniform candidates, a candidate is dropped when an earlier one
    lies within 0.55 of the mean spacing ``1/n`` - Gmsh's nodes keep their distance too), ``density`` x the structured
    beam's node count in all; on the clamp plane ``x = 0`` the exact grid of spacing ``1/n`` (planar, so that
    ``Data_prepare.py:127-136`` finds its facets); on the other five faces the grid points moved by up to 0.3/n inside the
    face and lifted outward by <= 0.03/n along a smooth bulge (a hull in general position: with coplanar face points Qhull
    needs 8x as long; the shape changes by 0.2 % of the width).  Tets are oriented positively (the reference keeps ``detJ``
    signed, ``Mat_construction.py:93``); slivers - volume below ``sliver`` x the cube of the longest edge, whose
    ``1/detJ`` makes any two fp64 evaluations of ``K_e`` disagree and which no explicit scheme with an edge-based time
    step survives - are dropped (10 % of the tets, 4 % of the volume: small voids).  Deterministic for a given ``seed``.
    n = 19: 190 400 nodes (the structured beam's count: one resident workgroup per CU still holds it), 1 002 482 tets,
    10-20 s."""
    from scipy.spatial import Delaunay, cKDTree

    if n < 2:
        raise ValueError("n must be >= 2")
    dims = np.array([length, width, height], dtype=np.float64)
    nx, ny, nz = int(round(length / width)) * n, n, n
    h = width / n
    X, Y, Z = np.meshgrid(np.linspace(0.0, length, nx + 1), np.linspace(0.0, width, ny + 1),
                          np.linspace(0.0, height, nz + 1), indexing="ij")
    on_face = np.zeros(X.shape, dtype=bool)
    on_face[[0, -1], :, :] = on_face[:, [0, -1], :] = on_face[:, :, [0, -1]] = True
    grid = np.stack([X[on_face], Y[on_face], Z[on_face]], axis=1)
    rng = np.random.default_rng(seed)
    wall = grid[:, 0] < 1e-12
    faces = grid.copy()
    for ax in range(3):  # inside the face
        free = (grid[:, ax] > 1e-9) & (grid[:, ax] < dims[ax] - 1e-9) & ~wall
        faces[free, ax] += rng.uniform(-0.3 * h, 0.3 * h, size=int(free.sum()))
    u = faces / dims
    for ax, (a, b) in enumerate(((1, 2), (0, 2), (0, 1))):  # outward, by a strictly concave function of the other two
        bump = 0.02 * h * (2.0 * (u[:, a] * (1.0 - u[:, a]) + u[:, b] * (1.0 - u[:, b])) + 0.05)
        lo = (grid[:, ax] < 1e-9) & ~wall
        hi = (grid[:, ax] > dims[ax] - 1e-9) & ~wall
        if ax > 0:
            faces[lo, ax] -= bump[lo]
        faces[hi, ax] += bump[hi]
    want = int(density * X.size) - len(faces)
    cand = rng.uniform(0.5 * h, dims - 0.5 * h, size=(int(2.4 * want), 3))
    close = cKDTree(cand).query_pairs(0.55 * h, output_type="ndarray")  # pairs (i < j)
    drop = np.zeros(len(cand), dtype=bool)
    drop[close[:, 1]] = True  # a candidate with an earlier one too close
    pts = np.concatenate([faces, cand[~drop][:want]])
    tri = Delaunay(pts)
    tets = tri.simplices.astype(np.int64)
    p = pts[tets]
    e = p[:, 1:] - p[:, :1]
    vol = np.einsum("ij,ij->i", e[:, 0], np.cross(e[:, 1], e[:, 2])) / 6.0
    neg = vol < 0
    tets[neg] = tets[neg][:, [0, 1, 3, 2]]
    longest = np.zeros(len(tets))
    for a in range(4):
        for b in range(a + 1, 4):
            longest = np.maximum(longest, np.linalg.norm(p[:, a] - p[:, b], axis=1))
    tets = tets[np.abs(vol) > sliver * longest ** 3]
    used = np.unique(tets)
    remap = np.full(len(pts), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    hull = tri.convex_hull
    hull = hull[np.all(pts[hull, 0] < 1e-12, axis=1) & np.all(remap[hull] >= 0, axis=1)]
    return Mesh(pts[used], {"tetra": remap[tets], "triangle": remap[hull]})


def clamp_nodes(mesh: Mesh, tol: float = 1e-9) -> np.ndarray:
    """Nodes of facets lying on ``x = 0`` in first-seen order (``Data_prepare.py:127-136``)."""
    tri = mesh.triangles
    if tri.size == 0:
        return np.zeros(0, dtype=np.int64)
    on = np.all(np.abs(mesh.points[tri, 0]) < tol, axis=1)
    flat = tri[on].ravel()
    _, first = np.unique(flat, return_index=True)
    return flat[np.sort(first)]


def elmdist(n_elems: int, size: int) -> np.ndarray:
    """Contiguous element ranges handed to ParMETIS (``Data_prepare.py:66-71``)."""
    n_each = n_elems // size
    n_left = n_elems - n_each * size
    head = (n_each + 1) * np.arange(n_left + 1)
    tail = (n_each + 1) * n_left + n_each * np.arange(1, size - n_left + 1)
    return np.append(head, tail).astype(np.int64)


def slab_partition(mesh: Mesh, n_parts: int, axis: int = 0) -> np.ndarray:
    """Element -> part vector: ``n_parts`` slabs of (nearly) equal element count along ``axis``.

    Stands in for ``part_mesh_kway``'s ``epart`` (``Data_prepare.py:94-101``); for the beam
    it yields ``n_parts - 1`` planar interfaces, each shared by exactly two parts.
    """
    cent = mesh.points[mesh.tets, axis].mean(axis=1)
    order = np.argsort(cent, kind="stable")
    epart = np.empty(len(cent), dtype=np.int64)
    bounds = elmdist(len(cent), n_parts)
    # Layered meshes (few distinct node coordinates along the axis, e.g. the synthetic beams): move every cut to the
    # nearest node plane, so that an interface is one plane of nodes instead of a ragged band two layers thick
    # (twice the shared nodes for a 0.4 % better element balance).
    planes = np.unique(mesh.points[:, axis])
    if 2 < len(planes) <= 8192 and len(planes) * 4 < len(mesh.points):
        sorted_cent = cent[order]
        for r in range(1, n_parts):
            k = int(bounds[r])
            if 0 < k < len(cent):
                cut = 0.5 * (sorted_cent[k - 1] + sorted_cent[k])
                plane = planes[np.argmin(np.abs(planes - cut))]
                bounds[r] = int(np.searchsorted(sorted_cent, plane, side="left"))
        for r in range(1, n_parts):  # keep the slabs non-empty and ordered
            bounds[r] = min(max(bounds[r], bounds[r - 1] + 1), len(cent) - (n_parts - r))
    for r in range(n_parts):
        epart[order[bounds[r]: bounds[r + 1]]] = r
    return epart


def graph_partition(mesh: Mesh, n_parts: int, return_stats: bool = False):
    """Element -> part vector from the library's graph partitioner (``saa_part_mesh_kway``: recursive bisection of the
    dual graph, greedy growing + FM refinement) - the METIS-quality stand-in for ``part_mesh_kway``
    (``Data_prepare.py:82-101``) on meshes where slabs are poor.  Deterministic: every rank computes the same vector."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    tets = np.ascontiguousarray(mesh.tets, dtype=np.int32)
    epart = np.zeros(len(tets), dtype=np.int32)
    st = _lib.PartitionStats()
    ip = C.POINTER(C.c_int32)
    _lib.check(lib.saa_part_mesh_kway(int(n_parts), len(tets), len(mesh.points), tets.ctypes.data_as(ip),
                                      epart.ctypes.data_as(ip), C.byref(st)))
    epart = epart.astype(np.int64)
    return (epart, st.as_dict()) if return_stats else epart


def rcb_partition(mesh: Mesh, n_parts: int) -> np.ndarray:
    """Recursive coordinate bisection of element centroids into ``n_parts`` parts."""
    cent = mesh.points[mesh.tets].mean(axis=1)
    epart = np.zeros(len(cent), dtype=np.int64)

    def split(idx, first, count):
        if count == 1:
            epart[idx] = first
            return
        left = count // 2
        ext = cent[idx].max(axis=0) - cent[idx].min(axis=0)
        ax = int(np.argmax(ext))
        k = (len(idx) * left) // count
        part = np.argpartition(cent[idx, ax], k) if 0 < k < len(idx) else np.arange(len(idx))
        split(idx[part[:k]], first, left)
        split(idx[part[k:]], first + left, count - left)

    split(np.arange(len(cent)), 0, n_parts)
    return epart
