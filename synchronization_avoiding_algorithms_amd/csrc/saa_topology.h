// Partition bookkeeping of one rank on the GPU (saa_topology.hip): what Data_prepare.py:104-144 derives from the element
// partition with O(N^2) list scans (Tools/Distributed_tools.py:14-73), in the reference's orderings.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace saa {

struct RankTopology {
  std::vector<int32_t> elements;         // this rank's elements, ascending (rankwise_dist, Distributed_tools.py:14-24)
  std::vector<int32_t> nodes;            // its nodes in first-touch order (ibid.)
  std::vector<int32_t> cells_local;      // (n_elements, 4) local node ids (local_mat_node, :66-73)
  std::vector<int32_t> shared_nodes;     // its nodes held by other ranks too, in find_shared_nodes' order (:29-40)
  std::vector<int32_t> shared_local;     // their local ids
  std::vector<int32_t> shared_slots;     // their positions in global_shared
  std::vector<int32_t> global_shared;    // sorted union over all ranks (sort_shared, :44-51)
  std::vector<int32_t> dirichlet_nodes;  // nodes of facets on x = 0, first-seen order (Data_prepare.py:127-136)
  std::vector<int32_t> dirichlet_local;  // local ids of this rank's clamped nodes, ascending (Dirichlet_rank_dist, :55-62)
};

// Host arrays in, host vectors out; works on `device`, null stream.  `facets` (n_facets x 3 node ids) and `xyz` may be
// null: no Dirichlet detection then.  `err` is filled when the input is at fault (hipErrorInvalidValue).
hipError_t rank_topology(int device, int32_t n_nodes, int32_t n_elems, const int32_t *tets, const int32_t *epart, int32_t rank,
                         int32_t n_parts, const double *xyz, int32_t n_facets, const int32_t *facets, double clamp_tol,
                         RankTopology *out, std::string &err);

}  // namespace saa
