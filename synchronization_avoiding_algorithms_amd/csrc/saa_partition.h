// k-way element partition on the dual graph of a tetrahedral mesh (see saa_partition.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace saa {

struct PartitionStats {
  int64_t face_cut = 0;         // faces between elements of different parts
  int64_t min_part = 0, max_part = 0;  // elements in the smallest / largest part
  int32_t interface_nodes = 0;  // nodes touched by elements of more than one part (= len(Global_shared))
};

// epart[e] in [0, n_parts); deterministic.  False + err on bad input.
bool partition_kway(int32_t n_parts, int32_t n_elems, int32_t n_nodes, const int32_t *tets, std::vector<int32_t> &epart,
                    PartitionStats &stats, std::string &err);

}  // namespace saa
