// Block decomposition ("plan") of one mesh partition for the fused explicit-step kernel.
//
// Owner-computes layout: nodes are renumbered so that every workgroup owns one CONTIGUOUS range
// of nodes (a compact box from recursive coordinate bisection).  A block's element list holds every
// element touching at least one owned node (elements on block borders are duplicated, never summed
// across workgroups), with connectivity stored as 16-bit block-local indices: owned nodes first,
// then the block's halo nodes.  Result: f_int needs no global atomics and no second pass, and its
// summation order is fixed by the plan.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace saa {

struct BlockDesc {
  int32_t node_start;  // first owned node (internal numbering)
  int32_t n_owned;
  int32_t halo_off;    // offset into halo_ids
  int32_t n_halo;
  int32_t elem_off;    // offset into conn (element copies)
  int32_t n_elem;
  int32_t n_interior;  // elements [0,n_interior) touch owned nodes only; the rest need halo records
  int32_t pad_;
};

struct Plan {
  int32_t n_nodes = 0, n_elems = 0;
  std::vector<int32_t> new_to_old;  // internal node id -> caller's node id
  std::vector<int32_t> old_to_new;
  std::vector<BlockDesc> blocks;
  std::vector<int32_t> halo_ids;    // internal node ids, per block sorted ascending
  std::vector<uint16_t> conn;       // 4 block-local indices per element copy
  int32_t max_owned = 0, max_local = 0;
  int64_t n_elem_copies = 0, n_halo_total = 0;
};

// Largest number of block-local nodes (owned + halo) a workgroup may stage; bounded by the 16-bit
// local indices and by the LDS budget the kernels are compiled for.
constexpr int32_t kMaxLocalNodes = 2560;  // 2560 * 48 B = 120 KiB of node records
constexpr int32_t kDefaultBlockNodes = 384;

// Builds the plan; on failure returns false and fills err.  block_nodes <= 0 selects the default.
bool build_plan(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err);

}  // namespace saa
