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
  int32_t owned_limit; // lds_index(n_owned): connectivity entries below it refer to owned nodes
};

// LDS image of a block: tiles of kTileNodes nodes, each tile 9 planes (x y z ux uy uz fx fy fz) of
// kTileNodes doubles.  Plane stride = 288 doubles = 2304 B = 9 * 256 B: (a) a multiple of the 256-B
// bank row, so every plane of a node sits in the same LDS bank pair (slot mod 32) and ONE ordering
// criterion makes record reads and force atomics conflict-free; (b) beyond the reach of
// ds_read2_b64 (8-bit offsets of 8 B) and not a multiple of 512 B (ds_read2st64_b64), so hipcc keeps
// the full-rate ds_read_b64 instead of pairing planes into half-rate two-address reads.
constexpr int32_t kTileNodes = 288;
constexpr int32_t kTilePlanes = 9;
constexpr int32_t kTileDoubles = kTileNodes * kTilePlanes;  // 2592 (= 81 * 32: tiles keep the bank phase)
inline int32_t lds_index(int32_t local_node) {
  return (local_node / kTileNodes) * kTileDoubles + (local_node % kTileNodes);
}
inline int32_t lds_bytes_for(int32_t max_local) {
  return ((max_local + kTileNodes - 1) / kTileNodes) * kTileDoubles * 8;
}

struct Plan {
  int32_t n_nodes = 0, n_elems = 0;
  std::vector<int32_t> new_to_old;  // internal node id -> caller's node id
  std::vector<int32_t> old_to_new;
  std::vector<BlockDesc> blocks;
  std::vector<int32_t> halo_ids;    // internal node ids, per block sorted ascending
  std::vector<uint16_t> conn;       // 4 LDS indices (lds_index of the block-local node) per element copy
  double lds_conflict_factor = 1.0; // mean over (half-wave, vertex slot) of the worst bank multiplicity
  int32_t max_owned = 0, max_local = 0;
  int64_t n_elem_copies = 0, n_halo_total = 0;
};

// Largest number of block-local nodes (owned + halo) a workgroup may stage; bounded by the 16-bit
// local indices and by the LDS budget the kernels are compiled for.
constexpr int32_t kMaxLocalNodes = 2016;  // 7 tiles * 20.25 KiB = 141.75 KiB of LDS
constexpr int32_t kDefaultBlockNodes = 384;

// Builds the plan; on failure returns false and fills err.  block_nodes <= 0 selects the default.
bool build_plan(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err);

}  // namespace saa
