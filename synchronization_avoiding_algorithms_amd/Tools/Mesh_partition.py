"""Stand-in for ``mgmetis.parmetis.part_mesh_kway`` as the reference calls it (``Data_prepare.py:82-101``,
``Online_predictor.py:95-114``): ``_, epart = part_mesh_kway(size, eptr, eind)`` with this rank's contiguous slice of
the element list (``elmdist``) in CSR form, returning the part of each element of the slice.

ParMETIS partitions the distributed slices collectively over MPI.  Here every rank holds the whole mesh anyway
(``Data_prepare.py:76-79``), so the slices are put together (``torch.distributed`` all-gather when a process group with
more than one rank exists), the library's deterministic graph partitioner (``saa_part_mesh_kway``: recursive bisection
of the dual graph, BFS growing + Fiduccia-Mattheyses refinement) runs redundantly on every rank, and each rank keeps the
entries of its slice - no partition data travels afterwards."""
from __future__ import annotations

import numpy as np


def part_mesh_kway(nparts, eptr, eind, group=None):
    """Returns ``(objval, epart)``: the part of every element of this rank's slice and, as ``objval``, the number of
    element FACES cut by the partition (edges of the dual graph between parts).  METIS' ``objval`` is its own objective
    (edge cut or communication volume of the graph it built from ``ncommon``): the two numbers are not comparable, and
    the reference discards the value (``_, epart = ...``).  Fails when a part would come out empty."""
    import ctypes as C

    from .. import _lib

    eptr = np.asarray(eptr, dtype=np.int64)
    eind = np.asarray(eind, dtype=np.int64)
    if np.any(np.diff(eptr) != 4):
        raise NotImplementedError("4-node tetrahedra only (the explicit path, Data_prepare.py:43-44)")
    mine = eind.reshape(-1, 4)
    slices, rank = [mine], 0
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            rank = dist.get_rank(group)
            slices = [None] * dist.get_world_size(group)
            dist.all_gather_object(slices, mine, group=group)
    except ImportError:  # pragma: no cover - torch is a hard dependency of the package
        pass
    tets = np.ascontiguousarray(np.concatenate(slices), dtype=np.int32)
    start = sum(len(s) for s in slices[:rank])
    epart = np.zeros(len(tets), dtype=np.int32)
    st = _lib.PartitionStats()
    ip = C.POINTER(C.c_int32)
    _lib.check(_lib.load().saa_part_mesh_kway(int(nparts), len(tets), int(tets.max()) + 1 if len(tets) else 1,
                                              tets.ctypes.data_as(ip), epart.ctypes.data_as(ip), C.byref(st)))
    return int(st.face_cut), epart[start:start + len(mine)].astype(np.int64)
