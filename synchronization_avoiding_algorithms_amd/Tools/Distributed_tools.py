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
    entered by every rank, like the reference's.  The reference's root adds the ranks' vectors in RANK order
    (``:84-86``); an all-reduce adds them in whatever order its ring takes, which gives other last bits wherever three
    or more ranks hold a node - those dofs (none on slab partitions; edges and corners of a k-way partition) are gathered
    and summed again in rank order, so that the result has the reference's bits on every rank.  ``f`` may be NumPy
    ``(3n,1)`` (returns NumPy) or a torch tensor on the process group's device (returns a tensor)."""
    import torch
    import torch.distributed as dist

    dof = node_to_dof(3, [0, 1, 2], Local_nodes)

    def reduce_in_rank_order(g):
        # the holder count of every node rides along in the same all-reduce (L_g more values): which dofs have three or
        # more holders is then known to every rank from the SAME reduced data in the SAME call - no cache, no second
        # collective whose entry could depend on a rank's history
        local = g.clone()
        idx_nodes = torch.as_tensor(np.asarray(Local_nodes, dtype=np.int64), device=g.device)
        buf = torch.cat((g.reshape(-1), torch.zeros(int(L_g), dtype=g.dtype, device=g.device)))
        buf[3 * int(L_g) + idx_nodes] = 1.0
        dist.all_reduce(buf, group=group)
        g = buf[:3 * int(L_g)].reshape(-1, 1).clone()
        multi_nodes = torch.nonzero(buf[3 * int(L_g):] >= 3).reshape(-1)  # (small integers: exact in floating point)
        if multi_nodes.numel() > 0:  # the same decision on every rank
            multi = (3 * multi_nodes[:, None] + torch.arange(3, device=g.device)[None, :]).reshape(-1)
            parts = [torch.empty((multi.numel(), 1), dtype=g.dtype, device=g.device) for _ in range(size)]
            dist.all_gather(parts, local[multi].contiguous(), group=group)
            acc = torch.zeros_like(parts[0])
            for p in parts:  # f_global[...] += f of rank 0, 1, 2, ... (ranks without the node add an exact zero)
                acc = acc + p
            g[multi] = acc
        return g

    if isinstance(f, torch.Tensor):
        g = torch.zeros((3 * L_g, 1), dtype=f.dtype, device=f.device)
        idx = torch.as_tensor(dof, device=f.device)
        g[idx] = f.reshape(-1, 1)
        if size != 1:
            g = reduce_in_rank_order(g)
        return g[idx]
    g = torch.zeros((3 * L_g, 1), dtype=torch.float64)
    g[dof] = torch.from_numpy(np.ascontiguousarray(f, dtype=np.float64).reshape(-1, 1))
    if size != 1:
        backend = dist.get_backend(group)
        if backend == "nccl":
            g = reduce_in_rank_order(g.cuda()).cpu()
        else:
            g = reduce_in_rank_order(g)
    return g.numpy()[dof]
