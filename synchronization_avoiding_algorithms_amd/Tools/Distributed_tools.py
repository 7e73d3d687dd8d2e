"""Counterpart of the reference's ``Tools/Distributed_tools.py``: partition bookkeeping with the same
orderings, and ``syn_cpus`` as ONE all-reduce (``torch.distributed``: RCCL on GPUs, gloo on CPUs)."""
from __future__ import annotations

import numpy as np

from ..fem_setup import first_touch_nodes
from .commons import node_to_dof


def rankwise_dist(rank, recvbuf, Points, Cells):
    """Elements of ``rank`` and its nodes in first-touch order (``Distributed_tools.py:14-24``)."""
    ele = np.nonzero(np.asarray(recvbuf) == rank)[0]
    return ele.tolist(), first_touch_nodes(np.asarray(Cells)[ele]).tolist()


def find_shared_nodes(rank, size, rank_nodal_num, rank_nodal_list):
    """``Distributed_tools.py:29-40``: my nodes that appear on other ranks, in their list order."""
    mine = np.asarray(rank_nodal_list[rank], dtype=np.int64)
    hits = [np.asarray(rank_nodal_list[r], dtype=np.int64) for r in range(size) if r != rank]
    hits = [h[np.isin(h, mine)] for h in hits]
    if not hits or sum(len(h) for h in hits) == 0:
        return []
    cat = np.concatenate(hits)
    _, first = np.unique(cat, return_index=True)
    return cat[np.sort(first)].tolist()


def sort_shared(G_shared_nodes):
    """Sorted union (``Distributed_tools.py:44-51``)."""
    parts = [np.asarray(s, dtype=np.int64) for s in G_shared_nodes if len(s)]
    return np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int64)


def Dirichlet_rank_dist(D_node, Local_N_list):
    """Local dofs of clamped nodes (``Distributed_tools.py:55-62``)."""
    loc = np.nonzero(np.isin(np.asarray(Local_N_list), np.asarray(D_node)))[0]
    return node_to_dof(3, [0, 1, 2], loc)


def local_mat_node(G_ID, L_N):
    """Global -> rank-local node ids (``Distributed_tools.py:66-73``) in O(n log n)."""
    L_N = np.asarray(L_N, dtype=np.int64)
    order = np.argsort(L_N, kind="stable")
    pos = np.searchsorted(L_N[order], np.asarray(G_ID, dtype=np.int64))
    return order[pos].tolist()


def syn_cpus(size, rank, f, L_g, Local_nodes, group=None):
    """Sum of every rank's local force vector on the global numbering, restricted back
    (``Distributed_tools.py:77-92``).  One ``all_reduce`` instead of gather + root add + bcast; must be
    entered by every rank, like the reference's.  ``f`` may be NumPy ``(3n,1)`` (returns NumPy) or a torch
    tensor on the process group's device (returns a tensor)."""
    import torch
    import torch.distributed as dist

    dof = node_to_dof(3, [0, 1, 2], Local_nodes)
    if isinstance(f, torch.Tensor):
        g = torch.zeros((3 * L_g, 1), dtype=f.dtype, device=f.device)
        idx = torch.as_tensor(dof, device=f.device)
        g[idx] = f.reshape(-1, 1)
        if size != 1:
            dist.all_reduce(g, group=group)
        return g[idx]
    g = torch.zeros((3 * L_g, 1), dtype=torch.float64)
    g[dof] = torch.from_numpy(np.ascontiguousarray(f, dtype=np.float64).reshape(-1, 1))
    if size != 1:
        backend = dist.get_backend(group)
        if backend == "nccl":
            gd = g.cuda()
            dist.all_reduce(gd, group=group)
            g = gd.cpu()
        else:
            dist.all_reduce(g, group=group)
    return g.numpy()[dof]
