// Structures shared between the kernels (saa_kernels.hip) and the C-ABI layer (saa_api.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "saa_plan.h"

namespace saa {

// node tag: bits 0..2 = Dirichlet mask per component, bit 3 = shared node, bits 8.. = interface slot
constexpr int32_t kTagShared = 1 << 3;
constexpr int kTagSlotShift = 8;

struct DeviceMesh {
  const BlockDesc *blocks;
  const int32_t *halo_ids;
  const uint2 *conn;   // work items packed in 64 bits: 5 x 12-bit node index + 2-bit flag, see saa_plan.h
  const double *xyz;   // (n_nodes,3) internal order
  const double *mass;  // (3 n_nodes)
  const double *mass_node;  // (n_nodes) when every node has one mass for its three dofs (the reference's
                            // lumped mass always does), else nullptr: saves 16 B per node and step
  const double *fext;  // (3 n_nodes) un-ramped
  const double *fext_yz;    // (n_nodes) when the load is (0, v, v) on every node - the reference's body force
                            // (0,-fz,-fz)*V/4, commons.py:35-41 - else nullptr: saves 16 B per node and step
  const int32_t *tag;  // (n_nodes)
  const int32_t *slot_sidx;  // (n_global_shared) interface slot -> index in the caller's shared list, -1 if foreign
  double lambda6, mu6;  // Lame parameters / 6: the factor of the element volume detJ/6 (tet_core)
  int32_t n_blocks, n_nodes, max_local, max_owned;
};

// Scalars of one step, pre-computed on the host exactly as Python evaluates them
// (Dynamic_solver.py:13,17): dt2 = dt**2, half_dt = dt/2, half_alpha = 0.5*alpha,
// ramp = linear_ramp(tn).
struct StepConsts {
  double dt, dt2, half_dt, alpha, half_alpha, ramp;
};

// Deterministic mode: per-item force vectors and, per owned node (internal order), the list of vectors addressed to it.
struct DetLists {
  double *item_force;          // (n_items, 5 slots a p q r b, 3)
  const int64_t *contrib_off;  // (n_nodes + 1)
  const int32_t *contrib;      // global item index * 8 + slot, ascending
};

struct SharedMap {
  const int32_t *node;          // (n_shared) internal node id, caller's shared order
  const int32_t *slot;          // (n_shared) interface slot
  const int32_t *foreign_slot;  // (n_foreign) interface slots of shared nodes held by other ranks only
  int32_t n_shared, n_foreign;
};

// Direct peer exchange of the shared-node forces (saa_peer_attach): every rank owns an "inbox" in fine-grained
// device memory that its neighbours (ranks holding at least one common shared node) write over xGMI, inside the
// fused step kernel.  Low-latency protocol: every fp64 value travels as ONE 16-byte entry of two 8-byte words, each
// carrying 32 bits of the value and the 32-bit sequence number of the step, so data and "ready" flag arrive together
// and every word validates itself; the receiver polls the entry - no fence, no separate flag, no collective.
//   inbox (PeerEntry): [parity 0|1][sender rank 0..world-1][3 * n_shared of the OWNER]; within a sender's segment the
//   entries follow the SENDER's push order (its common nodes in internal-node order), so that the 64 lanes of a
//   pushing wave write 1 KiB of contiguous remote memory (full xGMI packets instead of 8-byte ones).
struct PeerEntry {
  unsigned long long lo, hi;  // (seq << 32) | low / high half of the double
};
// One 16-byte load each gives a pushing / collecting lane everything it needs for a node with ONE other holder
// (a dependent chain of index loads would sit on the kernel's tail); further holders use the generic lists.
struct PeerPushRec {
  PeerEntry *dst0;     // first neighbour: remote address of component 0, parity 0
  int32_t pstride0;    // entries between that neighbour's parity-0 and parity-1 inbox
  int32_t info;        // bits 0..15 node index inside its plan block, bits 16..23 number of neighbours, bit 30: this
                       // rank is the highest-ranked holder of the node (kPeerInfoHighest)
};
constexpr int32_t kPeerInfoHighest = 1 << 30;
struct PeerRecvRec {
  unsigned long long holders;  // bit p set: rank p holds that node
  int32_t recv0;               // first neighbour: entry index of component 0 in this rank's parity-0 inbox
  int32_t sidx;                // position in the caller's shared list
};
// A node's SECOND other holder (the edges of a k-way partition: three ranks on a node), for the resident kernel, which
// keeps these records in LDS; four or more holders take the generic lists.
struct PeerSecondRec {
  PeerEntry *dst1;   // second neighbour: remote address of component 0, parity 0 (nullptr: none)
  int32_t pstride1;  // entries between that neighbour's parity-0 and parity-1 inbox
  int32_t recv1;     // entry index of component 0 in this rank's parity-0 inbox
};
struct PeerMap {
  const PeerPushRec *push_rec;        // (n_shared)
  const PeerRecvRec *recv_rec;        // (n_shared)
  const PeerSecondRec *second_rec;    // (n_shared)
  const int32_t *blk_off;             // (n_blocks + 1) plan block -> range in the node-sorted shared list
  const int32_t *node;                // (n_shared) internal node id, ascending
  const int32_t *sidx;                // (n_shared) position in the caller's shared list
  const unsigned long long *holders;  // (n_shared) bit p set: rank p holds that node
  const int32_t *nb_off;              // (n_shared + 1) range of neighbour entries of that node (other holders, by rank)
  PeerEntry *const *push_dst;         // (n_nb) remote address of component 0 in that neighbour's parity-0 inbox
  const int64_t *push_pstride;        // (n_nb) entries between that neighbour's parity-0 and parity-1 inbox
  const int64_t *recv_idx;            // (n_nb) entry index of component 0 in THIS rank's parity-0 inbox
  const PeerEntry *inbox;             // this rank's inbox
  int64_t parity_stride;              // entries between this rank's parity-0 and parity-1 inbox
  int32_t *err;                       // set to 1 when a wait timed out
  int64_t timeout_ticks;              // wall_clock64() ticks (100 MHz) before a wait gives up
  int32_t rank, world, n_shared;
};

void launch_fused_step_peer(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const double *d0,
                            const double *dn, double *d1, double *hist_row, const StepConsts &k,
                            const PeerMap *pm_dev, unsigned seq);
void launch_peer_selftest(const PeerMap &pm, hipStream_t st, const double *own, double *out, unsigned seq);
// Resident multi-step kernel (persistent_steps_kernel): ONE launch advances the partition by many time
// steps.  Everything static (coordinates, mass, load, tags, connectivity) and the block's own displacements d^n,
// d^(n-1) stay in LDS between steps; per step a workgroup only publishes its new displacements and re-reads those of
// its halo nodes, as self-validating stamped entries (PeerEntry) - no flags, no grid barrier.
struct PersistArgs {
  double *g0, *g1;          // at launch g0 = d^n, g1 = d^(n-1) (internal order); step s writes d^(n+s+1) alternately
  PeerEntry *entries;       // 2 x entry_stride stamped displacements (parity of the step count selects the half)
  int64_t entry_stride;     // 3 * n_nodes
  int32_t step_base;        // steps taken by earlier launches of this kernel (stamps continue from it)
  int32_t nsteps;
  double tn0;               // time of d^n at launch
  int32_t ramp_on;          // linear_ramp (commons.py:7-11) on / off
  int32_t max_items;        // LDS capacity for work items
  const double *table;      // predicted phase (row = table_row0 + step), or nullptr
  double *hist;             // history rows (hist_row0 + step), or nullptr
  int64_t table_row0, hist_row0, width;  // width = 3 * n_shared
  int32_t *err;             // set to 1 when a wait timed out
  int64_t timeout_ticks;
  const PeerMap *peer;      // device copy of the peer map (synchronised steps with the peer exchange), or nullptr
  int32_t peer_rec_off;     // PEER: byte offset in the LDS image of the block's copy of its push / receive records
  uint32_t peer_seq_base;   // sequence number of the last exchange before this launch
  // trajectory recorder (saa_set_recorder): d^(n+1) of step index i goes to column i / save_every of a row-major
  // (3*n_nodes, n_cols) matrix in the CALLER's dof order whenever i % save_every == 0 (Data_prepare.py:238-240)
  double *traj;             // or nullptr
  const int32_t *new_to_old;
  int64_t traj_cols, step_index0;  // step index of the first step of this launch
  int32_t save_every;
  StepConsts consts;        // the PEER variant reads the step constants from here once per step (update phase) instead
                            // of holding the by-value copy in scalar registers through the item loops
  int32_t *census;          // non-null: census launch - every workgroup checks in here and waits for the full count
};

// Recorder of the per-step paths: one small kernel after a step that is due.
void launch_record_column(int n_nodes, const int32_t *new_to_old, hipStream_t st, const double *d_internal, double *traj,
                          int64_t n_cols, int64_t col);
// LDS bytes the resident kernel needs for this plan (0 = it cannot hold it).
int persistent_lds_bytes(int max_local, int max_owned, int max_items, int max_halo);
// Dynamic-LDS limit of the PEER variant (its image is followed by the block's peer records).
hipError_t configure_persistent_peer(int lds_bytes);
// How many workgroups of the resident kernel can be co-resident on the device (0 on error).
int persistent_max_blocks(int device, int threads, int lds_bytes);
// Launches the resident kernel; `a` travels by value in the kernel-argument segment.
hipError_t launch_persistent_steps(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const StepConsts &k,
                                   const PersistArgs &a, int mode);

hipError_t configure_kernels(int lds_bytes);
hipError_t configure_det_kernels(int lds_bytes);
// One step (or, force_only, f = K d into `out`) without floating-point atomics: item kernel + node kernel.
void launch_det_step(const DeviceMesh &m, const DetLists &det, int threads, int lds_bytes, hipStream_t st, const double *d0,
                     const double *dn, double *out, double *iface, const double *table_row, double *hist_row,
                     const StepConsts &k, bool force_only);
// tn_dev (nullable): take the ramp from the time stored there instead of k.ramp (graph-replayed steps)
void launch_fused_step(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const double *d0,
                       const double *dn, double *d1, double *iface, const double *table_row, double *hist_row,
                       const StepConsts &k, const double *tn_dev = nullptr);
void launch_set_scalar(hipStream_t st, double *p, double v);
#ifdef SAA_DIAGNOSTICS
void launch_fused_step_ablated(int variant, const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st,
                               const double *d0, const double *dn, double *d1, const StepConsts &k, double *dbg);
#endif
void launch_force_only(const DeviceMesh &m, int threads, int lds_bytes, hipStream_t st, const double *d,
                       double *f);
// tn_in / tn_out (nullable pair): device-side clock of graph-replayed steps - ramp from *tn_in, *tn_out = *tn_in + dt
void launch_iface_finish(const DeviceMesh &m, const SharedMap &sh, hipStream_t st, const double *d0,
                         const double *dn, double *d1, double *iface, double *hist_row, const StepConsts &k,
                         const double *tn_in = nullptr, double *tn_out = nullptr);
void launch_halo_overwrite(const SharedMap &sh, hipStream_t st, const double *row, double *d1, double *hist_row);
void launch_halo_gather(const SharedMap &sh, hipStream_t st, const double *d, double *row);
void launch_cd_update(const DeviceMesh &m, hipStream_t st, const double *f_int, const double *d0,
                      const double *dn, double *d1, const StepConsts &k);
void launch_permute(int n_nodes, const int32_t *new_to_old, hipStream_t st, const double *in, double *out);
void launch_unpermute(int n_nodes, const int32_t *new_to_old, hipStream_t st, const double *in, double *out);

}  // namespace saa
