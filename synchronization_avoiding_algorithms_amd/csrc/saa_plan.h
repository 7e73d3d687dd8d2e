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
#include <cstdlib>
#include <string>
#include <vector>

namespace saa {

// Switches of measured-and-rejected plan / launch variants (SAA_PLAN_*, SAA_PERSIST_CHUNK, SAA_SHARED_NODE_WORK,
// SAA_RESIDENT_TRUST_GRID, SAA_NO_PERSISTENT) exist only in the diagnostic build (-DSAA_DIAGNOSTICS, tools/); the product
// library reads no environment variable.
#ifdef SAA_DIAGNOSTICS
inline const char *diag_env(const char *name) { return std::getenv(name); }
#else
inline const char *diag_env(const char *) { return nullptr; }
#endif

struct BlockDesc {
  int32_t node_start;  // first owned node (internal numbering)
  int32_t n_owned;
  int32_t halo_off;    // offset into halo_ids
  int32_t n_halo;
  int32_t elem_off;    // offset into conn, in work items
  int32_t n_elem;      // work items of the block (pairs of face-adjacent elements, or single elements)
  int32_t n_interior;  // items [0,n_interior) touch owned nodes only and may run before the halo records arrive; the rest
                       // wait for them.  (One block per CU, 1024 threads: at most 1024 - one round; further all-owned
                       // items are packed into the second list together with those that need halo records.)
  int32_t reserved;    // (keeps the descriptor at 32 bytes)
};

// LDS image of a block (doubles): node records [n_local][6] = x y z ux uy uz (48 B, 16-B aligned: three
// ds_read_b128 per node), then the force accumulators [n_owned][3] (AoS: one address computation per node, the
// components at immediate offsets).
//   * ds_read_b128 is executed per 16-lane group; lanes of a group collide unless their nodes differ mod 16
//     (record start bank = 12*node mod 64 takes 16 distinct values);
//   * ds_add_f64 is executed per 32-lane half; lanes collide unless their nodes differ mod 32 (accumulator start bank
//     = 6*node mod 64).
// The plan packs elements so that both hold as far as possible (reorder_for_lds).
inline int32_t lds_bytes_for(int32_t max_local, int32_t max_owned) {
  return (8 * (6 * max_local + 3 * max_owned) + 15) / 16 * 16;
}

struct Plan {
  int32_t n_nodes = 0, n_elems = 0;
  std::vector<int32_t> new_to_old;  // internal node id -> caller's node id
  std::vector<int32_t> old_to_new;
  std::vector<BlockDesc> blocks;
  std::vector<int32_t> halo_ids;    // internal node ids, per block (sorted until blocks renumber their nodes, see block_perm)
  // Work items, 8 x uint16 each: block-local node indices (a, p, q, r, b), flag, 0, 0.  flag = 1: the two
  // face-adjacent tets A = (a; p,q,r) and B = (b; p,r,q) (5 node records and 5 force flushes for two
  // elements instead of 8 + 8); flag = 0: the single tet (a, p, q, r), b = p as a harmless dummy;
  // flag = 2: null item (an idle lane left by the LDS packing).
  std::vector<uint16_t> conn;
  // The same items as the device reads them: 64 bits each, node indices in 12-bit fields (a at bit 0, p 12,
  // q 24, r 36, b 48), flag in bits 60-61 - 8 bytes of connectivity per TWO elements.
  std::vector<uint64_t> conn_packed;
  int64_t n_items = 0, n_pairs = 0;
  double lds_conflict_factor = 1.0; // mean over (ds_read_b128 lane group, vertex slot) of the worst bank multiplicity
  double lds_atomic_conflict_factor = 1.0;  // the same for the ds_add_f64 of a half-wave (owned vertices only)
  int64_t n_by_construction = 0;    // item slots in half-waves that are clash-free by construction (pattern classes)
  int32_t n_renumbered = 0;         // blocks whose nodes took another order than the plan's (axis order / pseudo-lattice)
  int32_t max_owned = 0, max_local = 0;
  int64_t n_elem_copies = 0, n_halo_total = 0;
};

// Largest number of block-local nodes (owned + halo) a workgroup may stage; bounded by the 16-bit
// local indices and by the LDS budget the kernels are compiled for.
constexpr int32_t kMaxLocalNodes = 2730;  // 2730 * 48 B = 128 KiB of node records; < 4096 (12-bit item fields)
constexpr int32_t kDefaultBlockNodes = 384;
constexpr int32_t kLargeMeshBlockNodes = 720;  // partitions too large for one block per CU

// Builds the plan; on failure returns false and fills err.  block_nodes <= 0 selects the default.
// extra_work (nullable, per node in the caller's numbering): work a node brings to its block besides its elements,
// in units of element copies (e.g. the peer exchange of a shared node); the blocks are balanced on the sum.
bool build_plan(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                int32_t block_nodes, Plan &plan, std::string &err, const int32_t *extra_work = nullptr);

}  // namespace saa
