#!/usr/bin/env python3
"""How many ranks hold a shared node?  The peer exchange keeps a node's first two other holders in 16-byte records (one
load each; in LDS in the resident kernel); nodes with four or more holders take the generic holder lists.  This counts
them on k-way GRAPH partitions (the library's own partitioner, the stand-in for ParMETIS) of the bench's beams, structured
and jittered/shuffled - no GPU needed.

    python tools/holder_census.py [n=19] [parts=8]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1] + sys.argv[1:]
from bench import bench_mesh  # noqa: E402
from synchronization_avoiding_algorithms_amd.mesh import graph_partition, slab_partition  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
from synchronization_avoiding_algorithms_amd.mesh import structured_beam  # noqa: E402

cases = [("structured", bench_mesh(n, "structured")), ("jittered", bench_mesh(n, "jittered")),
         ("cube", structured_beam(2 * n, length=1.0))]  # a compact domain: where a k-way partition has edges and corners
for kind, mesh in cases:
    for name, epart in (("graph partition", graph_partition(mesh, parts)), ("x-slabs", slab_partition(mesh, parts))):
        holders = np.zeros(len(mesh.points), dtype=np.int64)
        for p in range(parts):
            holders[np.unique(mesh.tets[epart == p])] += 1
        shared = holders[holders > 1]
        hist = np.bincount(shared, minlength=6)
        per_rank = []
        for p in range(parts):
            mine = np.unique(mesh.tets[epart == p])
            per_rank.append(int((holders[mine] >= 4).sum()))
        print(f"{kind:10s} n={n} {parts} parts, {name:15s}: {len(shared):6d} shared nodes; holders 2: {hist[2]}, 3: {hist[3]}, "
              f"4: {hist[4]}, >=5: {hist[5:].sum()}  ({100.0 * hist[4:].sum() / max(len(shared), 1):.2f} % with four or more; "
              f"most on one rank: {max(per_rank)})")
