/*
 * saa_hip.h - C ABI of libsaa_hip.so: the MI355X (gfx950) implementation of the explicit
 * linear-tetrahedral elastodynamics hot path of desResLab/Synchronization-avoiding-algorithms.
 *
 * The reference has no FFI: its hot path sits behind Python call signatures
 * (SURVEY.md section 8(b)).  Each entry point below names the reference interface it replaces
 * (file:line in /root/reference); INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SAA_E_* code otherwise; the message of the
 *     last failure on the calling thread is returned by saa_last_error().
 *   - "host" pointers are caller-owned CPU buffers, "dev" pointers caller-owned device buffers
 *     (e.g. torch tensors' data_ptr()).  The library never frees caller memory.
 *   - node / dof numbering at this boundary is ALWAYS the caller's (the rank-local first-touch
 *     numbering of Distributed_tools.py:14-24; dof = 3*node + component, commons.py:66-71).
 *     Internally nodes are renumbered block-wise; that never shows through the ABI.
 *   - one handle per GPU partition; calls on one handle must be serialised by the caller.  All
 *     device work is enqueued on the stream given to saa_set_stream (default: the null stream),
 *     so PyTorch-ROCm and RCCL work ordered on the same stream needs no extra synchronisation.
 *   - fp64 throughout (the reference is float64 NumPy).
 */
#ifndef SAA_HIP_H
#define SAA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAA_OK 0
#define SAA_E_ARG (-1)     /* bad argument (null pointer, index out of range, degenerate element) */
#define SAA_E_HIP (-2)     /* a HIP runtime call failed (no device, out of memory, launch failure) */
#define SAA_E_STATE (-3)   /* call sequence error (e.g. step_finish without step_begin) */
#define SAA_E_CAPACITY (-4) /* a node block does not fit the LDS budget even at the smallest size */

typedef struct saa_solver saa_solver;

/*
 * Everything one rank holds after the set-up of Data_prepare.py:104-209, in the caller's numbering.
 */
typedef struct saa_problem {
  int32_t n_nodes;               /* len(Local_nodal_list), Data_prepare.py:104 */
  int32_t n_elems;               /* len(Local_ele_list) */
  const double *xyz;             /* host (n_nodes,3): Points[Local_nodal_list] */
  const int32_t *tets;           /* host (n_elems,4): local node ids (local_mat_node of Cells rows,
                                    Mat_construction.py:139) */
  const double *lumped_mass;     /* host (3*n_nodes): l_M, Data_prepare.py:202 */
  const double *f_ext;           /* host (3*n_nodes): F_rankwise (un-ramped), Data_prepare.py:201 */
  const int32_t *dirichlet_dofs; /* host (n_dirichlet): Local_Dirichlet, Data_prepare.py:144 */
  int32_t n_dirichlet;
  const int32_t *shared_nodes;   /* host (n_shared): local ids of this rank's shared nodes, in the
                                    order of `shared_nodes` (Data_prepare.py:112); defines the column
                                    order 3*i+c of the LSTM input (Online_predictor.py:126-129) */
  const int32_t *shared_slots;   /* host (n_shared): position of each shared node in the sorted
                                    Global_shared list (Data_prepare.py:121-124) */
  int32_t n_shared;
  int32_t n_global_shared;       /* len(Global_shared); interface buffer holds 3*n_global_shared */
  double lambda_;                /* elasticity.lmd, commons.py:17 */
  double mu;                     /* elasticity.mu */
  double dt;                     /* min CFL step, Data_prepare.py:147-154 */
  double alpha;                  /* mass-proportional damping `Damp`, Data_prepare.py:41 */
  int32_t ramp;                  /* 1: F_ext = F_rankwise*min(t,1) (Dynamic_solver.py:13); 0: constant */
  int32_t device;                /* HIP device ordinal */
  int32_t block_nodes;           /* target owned nodes per workgroup; 0 = automatic */
  int32_t threads;               /* workgroup size (multiple of 64, <= 1024); 0 = automatic */
} saa_problem;

/* Statistics of the block decomposition (for DESIGN.md / bench.py roofline bookkeeping). */
typedef struct saa_plan_stats {
  int32_t n_blocks;
  int32_t max_owned;        /* owned nodes of the largest block */
  int32_t max_local;        /* owned + halo nodes of the largest block */
  int64_t n_elem_copies;    /* sum over blocks of elements touching the block (>= n_elems) */
  int64_t n_halo_total;     /* sum over blocks of halo nodes */
  int32_t lds_bytes;        /* dynamic LDS per workgroup */
  int32_t threads;          /* workgroup size in use */
  double lds_conflict_factor; /* mean worst LDS bank multiplicity of the record reads per (lane group, vertex slot); 1 = none */
  double lds_atomic_conflict_factor; /* the same for the force accumulation (ds_add_f64) per (half-wave, vertex slot) */
  int64_t n_items;          /* work items (pairs of face-adjacent elements, single elements, idle slots of the packing) */
  int64_t n_pairs;          /* items that hold two elements */
  int64_t n_by_construction; /* item slots in half-waves that are clash-free by construction (pattern classes) */
  int32_t n_renumbered;     /* blocks whose nodes took another order than the plan's first choice (axis order / pseudo-lattice) */
  int32_t reserved;
} saa_plan_stats;

const char *saa_last_error(void);
/* Library / ABI version; bumps when this header changes (2: peer exchange and resident-kernel entry points; 3: loop-back attach; 4: partitioner and set-up kernels; 5: deterministic mode; 6: copy-bandwidth aid; 7: saa_plan_stats grew; 8: saa_predictor_*, saa_topology_*; 9: saa_set_option, saa_plan_stats.n_renumbered; 10: saa_plan_host_check). */
int32_t saa_abi_version(void);

/* Element partition, one part per rank / GPU: the role of `_, epart = part_mesh_kway(size, eptr, eind)` (mgmetis /
 * ParMETIS, Data_prepare.py:82-101).  Graph partitioning of the dual graph (elements adjacent across a face): recursive
 * bisection by greedy graph growing + Fiduccia-Mattheyses refinement; deterministic, host only, no HIP call - every rank
 * computes the same vector from the replicated mesh (Data_prepare.py:76-79) instead of running a distributed partitioner.
 * tets: (n_elems,4) node ids in [0, n_nodes); epart_out: (n_elems) part of every element; stats_out may be NULL. */
typedef struct saa_partition_stats {
  int64_t face_cut;          /* faces between elements of different parts */
  int64_t min_part, max_part; /* elements in the smallest / largest part */
  int32_t interface_nodes;   /* nodes touched by more than one part = len(Global_shared), Data_prepare.py:121-124 */
} saa_partition_stats;
int saa_part_mesh_kway(int32_t n_parts, int32_t n_elems, int32_t n_nodes, const int32_t *tets, int32_t *epart_out,
                       saa_partition_stats *stats_out);

/* Set-up fields on the GPU, O(N), for the elements given (a rank passes the elements touching its own nodes):
 *   lumped_mass_out (3*n_nodes): row sums of the consistent mass = sum_e rho*V_e/4 per node, on its three dofs
 *                                (Global_Assembly_no_bc + lumping_to_vec: Mat_construction.py:199-231, commons.py:103-107,
 *                                Data_prepare.py:175-176);
 *   f_pre_out       (3*n_nodes): pre-assembled un-ramped body force sum_e (V_e/4)*(0,-fz,-fz) (same call, F_pre);
 *   min_edge_out    (scalar)   : shortest element edge; Meshsize = 2*min_edge/sqrt(24) (commons.py:79-90), from which
 *                                dt = gamma*Meshsize/sqrt(E/rho/(1-nu^2)) (Data_prepare.py:147).
 * Replaces the reference's dense (3N)^2 assembly on rank 0.  Signed volumes (detJ/6, Mat_construction.py:93); no
 * floating-point atomics: nodal sums run in ascending element order (deterministic).  Host pointers in and out, any
 * output may be NULL. */
int saa_setup_fields(int32_t device, int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets, double rho,
                     double fz, double *lumped_mass_out, double *f_pre_out, double *min_edge_out);

/* Build the device-resident solver for one partition.  Replaces, for this path,
 * Local_assembly_for_stiffness (Mat_construction.py:122-150: no matrix is ever assembled) plus the
 * per-step argument marshalling of parallel_explicit_solver_dis_pre (Dynamic_solver.py:9-10).
 * Initial state is d0 = dn = 0, tn = 0 (Data_prepare.py:171-172,215). */
int saa_create(const saa_problem *problem, saa_solver **out);
int saa_destroy(saa_solver *s);
int saa_plan_stats_get(const saa_solver *s, saa_plan_stats *out);

/* Host-only plan builder (no HIP call): same decomposition saa_create uses, for CPU tests. */
int saa_plan_host_stats(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                        int32_t block_nodes, saa_plan_stats *out);
/* Self-check of the block plan the library would build for this partition (host only, no GPU): the internal numbering is a
 * permutation; every work item names valid nodes, its tets are elements of the mesh in the mesh's orientation (the signed
 * detJ of Mat_construction.py:93), first-round items name owned nodes only; every element appears exactly once in every
 * block owning one of its nodes and nowhere else.  *violations_out = number of failed checks (0: the plan is what the step
 * kernels assume).  No reference counterpart (its LocalK is an assembled matrix); used by the CPU tests. */
int saa_plan_host_check(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets, int32_t block_nodes,
                        int64_t *violations_out);

/* All later work of this handle goes to `hip_stream` (a hipStream_t; NULL = null stream). */
int saa_set_stream(saa_solver *s, void *hip_stream);

/* Time_integration_displacement(tn, dt, d0, dn) (commons.py:47-55): d0 = d^n, dn = d^(n-1). */
int saa_set_state(saa_solver *s, const double *d0_host, const double *dn_host, double tn);
int saa_get_state(saa_solver *s, double *d0_host, double *dn_host, double *tn);
/* Same, into caller-owned DEVICE buffers of 3*n_nodes doubles (either may be NULL). */
int saa_get_state_device(saa_solver *s, double *d0_dev, double *dn_dev);
/* Replace the un-ramped external force / lumped mass (host, 3*n_nodes each; NULL = keep). */
int saa_set_loads(saa_solver *s, const double *f_ext_host, const double *lumped_mass_host);

/* f = K_local . d without K: backs `LocalK.dot(T.d0)` (Dynamic_solver.py:12).  Host in, host out,
 * 3*n_nodes doubles each. */
int saa_internal_force(saa_solver *s, const double *d_host, double *f_host);
/* Same with caller-owned DEVICE buffers (3*n_nodes doubles each, caller numbering), enqueued on the handle's stream:
 * the operator of iterative solvers that keep their vectors on the GPU - the matrix-free replacement of
 * `np.linalg.solve(K, F)` in Steady_Elasticity_solver (Tools/Steady_solvers.py:13-22, Data_prepare.py:163). */
int saa_internal_force_device(saa_solver *s, const double *d_dev, double *f_dev);

/* One damped central-difference update on the host-provided arrays, evaluated on the GPU in the
 * reference's association order (Dynamic_solver.py:13-20).  Backs the drop-in
 * parallel_explicit_solver_dis_pre when the caller owns the state. */
int saa_cd_update(saa_solver *s, const double *f_int_host, const double *d0_host, const double *dn_host,
                  double tn, double *d1_host);

/* nsteps explicit steps with no exchange: the serial case (size == 1) and the MODEL=True branch
 * of Dynamic_solver.py:22 without a halo overwrite.  State rotates (dn<-d0<-d1), tn += dt. */
int saa_step(saa_solver *s, int32_t nsteps);

/* Synchronised step, split around the caller's collective (replaces syn_cpus,
 * Distributed_tools.py:77-92, by an all-reduce over the compact interface buffer):
 *   saa_step_begin : interior + shared nodes get the local update; the partial K_r d of every
 *                    local shared node is written to iface_dev[3*slot+c]; slots of shared nodes
 *                    this rank does not hold stay 0.
 *   (caller: all-reduce(sum) of iface_dev over ranks, on the same stream)
 *   saa_step_finish: shared nodes are recomputed from the summed force (Dynamic_solver.py:26-32),
 *                    non-local slots are re-zeroed, state rotates, tn += dt.  If hist_dev != NULL the
 *                    new shared-dof values are also written to hist_dev[hist_row*3*n_shared + ...]
 *                    (Online_predictor.py:260). */
int saa_set_interface_buffer(saa_solver *s, double *iface_dev);
int saa_step_begin(saa_solver *s);
int saa_step_finish(saa_solver *s, double *hist_dev, int64_t hist_row);

/* Optional native exchange: the per-step all-reduce issued from C++ on the handle's stream through the
 * RCCL library that is already loaded in the process (e.g. PyTorch-ROCm's librccl.so, given by path), so that
 * a synchronised step costs three enqueues and no interpreter time.
 *   saa_comm_unique_id : rank 0 creates the 128-byte ncclUniqueId; the caller broadcasts it to every rank
 *   saa_comm_init      : every rank joins (ncclCommInitRank); needs saa_set_interface_buffer first
 *   saa_step_synced    : nsteps x (saa_step_begin, ncclAllReduce(sum, fp64) of the interface buffer,
 *                        saa_step_finish); history rows hist_row0 + k as in saa_step_finish
 * With world == 1 the collective is the identity (used by the single-GPU tests). */
int saa_comm_unique_id(const char *rccl_path, uint8_t id_out[128]);
int saa_comm_init(saa_solver *s, const char *rccl_path, const uint8_t id[128], int32_t rank, int32_t world);
int saa_step_synced(saa_solver *s, int32_t nsteps, double *hist_dev, int64_t hist_row0);

/* Direct peer exchange: the synchronised step without a collective and without a second kernel.  Inside the fused
 * step kernel the partial force of every shared node is stored straight into the memory of the other ranks holding
 * that node (fine-grained device memory mapped through HIP IPC; xGMI peer stores; every 16-byte entry carries the
 * step's sequence number, so it is its own "ready" flag) and every rank sums what it received in RANK ORDER - the
 * summation order of syn_cpus (Distributed_tools.py:84-86), so all ranks obtain identical bits.  Only ranks with a
 * common shared node talk to each other; the xGMI flight time hides under the update of the non-shared nodes.
 *   saa_peer_export   : allocates this rank's inbox and returns its 64-byte hipIpcMemHandle_t plus, per shared
 *                       node, its position in this rank's push order (order_out, n_shared values); the caller
 *                       all-gathers handles, device ordinals, shared_slots lists and push orders of every rank
 *                       (any transport: torch.distributed, MPI, files)
 *   saa_peer_attach   : maps the inboxes of the neighbours (ranks with a common slot) and builds the push / receive
 *                       lists; handles = world x 64 bytes, devices[world], slot_counts[world], slots / orders = the
 *                       ranks' lists concatenated in rank order (this rank's slots must equal saa_problem's)
 *   saa_peer_selftest : COLLECTIVE over the attached ranks: one exchange of known values; *ok = 1 iff every
 *                       sum arrived intact within the time limit.  The caller agrees on min(ok) over ranks
 *                       before relying on saa_step_peer, and otherwise keeps the all-reduce path
 *   saa_step_peer     : nsteps synchronised steps, ONE kernel launch each; history rows as in saa_step_finish
 * Waits inside the kernel are bounded (30 s, env SAA_PEER_TIMEOUT_S): a dead neighbour turns into SAA_E_STATE at
 * the next saa_synchronize / saa_get_state instead of a hang. */
int saa_peer_export(saa_solver *s, int32_t world, uint8_t handle_out[64], int32_t *order_out);
int saa_peer_attach(saa_solver *s, int32_t rank, int32_t world, const uint8_t *handles, const int32_t *devices,
                    const int32_t *slot_counts, const int32_t *slots, const int32_t *orders);
int saa_peer_selftest(saa_solver *s, int32_t *ok);
/* Single-GPU rehearsal of the peer exchange (instead of saa_peer_export + saa_peer_attach): this handle becomes rank 0
 * of `world` (2..8) ranks whose other members are imaginary - they hold exactly this rank's shared nodes and their
 * inbox segments live in this rank's own inbox, so every pushed value comes straight back.  saa_step_peer then runs the
 * complete push / stamp / poll / rank-ordered-sum path of a real multi-GPU step with local instead of xGMI latency, and
 * the force a shared node is updated with is exactly world x its local partial force (a + a + ... in rank order) - a
 * well-defined operator the parity tests reproduce on the CPU (tests/test_gpu_fullsize.py: the per-GPU workload of the
 * 8-GPU configuration checked on one GPU).  Needs n_shared > 0. */
int saa_peer_attach_loopback(saa_solver *s, int32_t world);
int saa_step_peer(saa_solver *s, int32_t nsteps, double *hist_dev, int64_t hist_row0);

/* nsteps sync-free steps of the predicted phase (Online_predictor.py:287-316): after each local
 * update the shared dofs are overwritten by row (table_row0 + k) of table_dev (row length
 * 3*n_shared, fp64) and recorded into row (hist_row0 + k) of hist_dev (may be NULL). */
int saa_step_predicted(saa_solver *s, int32_t nsteps, const double *table_dev, int64_t table_row0,
                       double *hist_dev, int64_t hist_row0);

/* Trajectory recorder: the ground-truth loop's `d1_save[:, counter] = d1` (Data_prepare.py:236-240; likewise
 * Online_predictor.py:316-318) without leaving the GPU.  traj_dev is a caller-owned DEVICE matrix, row-major
 * (3*n_nodes, n_cols) in the caller's dof order - the layout of the reference's `Displacement` dataset.  From now on
 * every step taken through this handle (any stepping entry point, resident kernel included) has a step index,
 * starting at next_step_index; the displacement d^(n+1) of step index i is written to column i / save_every
 * whenever i % save_every == 0 and that column exists.  traj_dev = NULL switches the recorder off. */
int saa_set_recorder(saa_solver *s, double *traj_dev, int64_t n_cols, int32_t save_every, int64_t next_step_index);

/* d_sol_shared[i,:] = d0[loc_dof_shared] (Online_predictor.py:260,301) / the reverse overwrite
 * (:298) on the CURRENT d0, for callers that drive single steps themselves. */
int saa_halo_gather(saa_solver *s, double *row_dev);
int saa_halo_scatter(saa_solver *s, const double *row_dev);

/* Whether saa_step / saa_step_predicted / saa_step_peer calls of >= 8 steps run through the resident multi-step
 * kernel (one launch per steps_per_launch steps, the partition's image kept in LDS between steps,
 * DESIGN.md section 4) and how much LDS a workgroup of it holds.  capable = 0: the plan does not fit or the device
 * cannot keep all workgroups co-resident; every step is then one launch of the fused kernel. */
int saa_resident_kernel_info(const saa_solver *s, int32_t *capable, int32_t *lds_bytes, int32_t *steps_per_launch);
/* enable = 0: keep this handle on one launch per step (for callers that know the device is shared with other
 * processes: workgroups of a resident kernel that wait for another process' kernel only advance by time-slicing). */
int saa_set_resident_kernel(saa_solver *s, int32_t enable);
/* Run-time options of a handle, by name (the library itself reads no environment variable):
 *   "synced_graph"    1 (default) / 0: saa_step_synced replays HIP graphs of three steps each / enqueues every kernel and
 *                     collective itself (the role of the per-step MPI calls of Distributed_tools.py:77-92);
 *   "split_stepping"  1 (default) / 0: partitions too large for the resident kernel (several rounds of workgroups per
 *                     launch) step their blocks as three sets on three streams, so that one set's launch boundaries hide
 *                     under another's work / one launch of all blocks per step;
 *   "wait_timeout_s"  bound, in seconds (default 30), of every in-kernel wait for another workgroup or rank (resident
 *                     kernel, peer exchange); a wait that gives up is reported as SAA_E_STATE by saa_synchronize.
 * The reference has no counterpart (an MPI rank that loses its peer hangs, Distributed_tools.py:77-92). */
int saa_set_option(saa_solver *s, const char *name, double value);

/* Deterministic mode.  The step kernels accumulate the element forces of a node with LDS floating-point atomics, whose
 * order is free: f_int - and with it the trajectory - differs in the last bits from run to run (the same class of
 * difference as the reference's own dependence on partition and node numbering).  enable = 1 switches
 * this handle to a two-kernel form of the step without atomics: every work item writes its force vectors to memory and every
 * node adds the vectors addressed to it in a fixed order.  Same arithmetic per element, bit-identical results from run to
 * run; several times slower (a verification mode).  Covers saa_step, saa_step_begin/finish, saa_step_synced,
 * saa_step_predicted and saa_internal_force*; saa_step_peer is refused while it is on. */
int saa_set_deterministic(saa_solver *s, int32_t enable);

/* Blocks until all work enqueued for this handle has finished. */
int saa_synchronize(saa_solver *s);

/* Measurement aid for bench.py: device-to-device copy rate of this GPU - bytes read + bytes written per second by a
 * 16-byte-per-lane copy kernel over two buffers of n_bytes each, `reps` timed launches - the practical HBM ceiling that
 * SURVEY.md section 8(d) asks to be reported next to the nominal peak.  No counterpart in the reference (which has no
 * timing code at all: BASELINE.md section 1). */
int saa_device_copy_bandwidth(int32_t device, int64_t n_bytes, int32_t reps, double *bytes_per_s);

/*
 * Partition bookkeeping of ONE rank on the GPU: what Data_prepare.py:104-144 derives from the element partition `epart`
 * (`recvbuf`, Data_prepare.py:97-101) through Tools/Distributed_tools.py - `rankwise_dist` (:14-24: the rank's elements
 * and its nodes in first-touch order), `find_shared_nodes` (:29-40) + `sort_shared` (:44-51: `shared_nodes`,
 * `Global_shared`), `local_mat_node` (:66-73: local connectivity), `Dirichlet_rank_dist` (:55-62) - and the clamp detection
 * of Data_prepare.py:127-136 (nodes of boundary facets with all |x| < tol), in the reference's orderings, as O(N) device
 * passes + radix sorts instead of O(N^2) list scans.  Host arrays in; the results are held by the handle until copied out.
 *   tets   (n_elems, 4) GLOBAL node ids of the whole mesh, epart (n_elems) part of every element, 0 <= rank < n_parts;
 *   xyz (n_nodes, 3) and facets (n_facets, 3) may be null / 0: no clamp detection then.
 * saa_topology_sizes fills sizes[6] = { elements, nodes, shared nodes of the rank, Global_shared, clamped nodes of the
 * mesh, clamped nodes of the rank }; saa_topology_get copies into caller buffers of those sizes (any pointer may be null):
 *   elements (ascending), nodes (first-touch order), cells_local (elements x 4, local ids), shared_nodes (global ids,
 *   find_shared_nodes' order), shared_local, shared_slots (position in Global_shared), global_shared (sorted),
 *   dirichlet_nodes (global ids, first-seen order over the facets), dirichlet_local (local ids, ascending).
 */
typedef struct saa_topology saa_topology;
int saa_topology_build(int32_t device, int32_t n_nodes, int32_t n_elems, const int32_t *tets, const int32_t *epart, int32_t rank,
                       int32_t n_parts, const double *xyz, int32_t n_facets, const int32_t *facets, double clamp_tol,
                       saa_topology **out);
int saa_topology_sizes(const saa_topology *t, int32_t *sizes);
int saa_topology_get(const saa_topology *t, int32_t *elements, int32_t *nodes, int32_t *cells_local, int32_t *shared_nodes,
                     int32_t *shared_local, int32_t *shared_slots, int32_t *global_shared, int32_t *dirichlet_nodes,
                     int32_t *dirichlet_local);
int saa_topology_destroy(saa_topology *t);

/*
 * Shared-node predictor: the per-rank LSTM encoder-decoder of Tools/DNN_tools.py:16-98 (2-layer bidirectional encoder of
 * width hidden_size, decoder LSTM of width 2*hidden_size + Linear) evaluated for all filter_size phase offsets of one
 * prediction window - what Tools/DNN_prediction.py:38-55 (`encoder_decoder_predictor`) computes with filter_size
 * sequential batch-1 passes on the CPU - as four launches (two f32 matrix-core GEMMs for the input projections, one
 * recurrence kernel, one output GEMM that scales back and writes the table).
 *
 * saa_predictor_create replaces `call_model` (DNN_prediction.py:18-34): `weights` are host pointers to the 22 fp32
 * tensors of the reference's state_dict in ITS order (Model_training.py:179-180 saves them):
 *   encoder.lstm_encoder.{weight_ih, weight_hh, bias_ih, bias_hh}_l0, the same four _l0_reverse, _l1, _l1_reverse,
 *   decoder.lstm_decoder.{weight_ih, weight_hh, bias_ih, bias_hh}_l0, decoder.fc.weight, decoder.fc.bias
 * (row-major, PyTorch's gate order i, f, g, o).  They are copied; the caller may free them after the call.
 * filter_size >= 2, hidden_size <= 128.
 */
typedef struct saa_predictor saa_predictor;
int saa_predictor_create(int32_t device, int32_t input_size, int32_t hidden_size, int32_t n_past, int32_t n_future,
                         int32_t filter_size, const float *const *weights, int32_t n_weights, saa_predictor **out);

/* `encoder_decoder_predictor(device, n, model, n_p, n_f, n_s, input_size, d_sol, scale_max, scale_min)`
 * (DNN_prediction.py:38-55) on device buffers: reads rows [n - n_past*filter_size, n) of the fp64 history `hist_dev`
 * (hist_rows x input_size, row stride ld_hist doubles: d_sol of Online_predictor.py:260,301), scales them to [-1, 0]
 * (DNN_tools.py:272-275), runs the model in fp32 and writes the (filter_size*n_future) x input_size fp64 table (row stride
 * ld_table) whose row k is the prediction for step n + k (`NF`, DNN_prediction.py:45,53-54; what saa_step_predicted
 * consumes).  Enqueued on `stream` (a hipStream_t; null = the null stream); returns without synchronising. */
int saa_predictor_predict(saa_predictor *p, const double *hist_dev, int64_t hist_rows, int64_t ld_hist, int64_t n,
                          double scale_max, double scale_min, double *table_dev, int64_t ld_table, void *stream);

int saa_predictor_destroy(saa_predictor *p);

/* The pointwise part of one LSTM step and its backward pass, for the training loop (`model_train`, DNN_tools.py:103-165,
 * whose decoder steps are `torch.nn.LSTM` calls, DNN_tools.py:73-79): device pointers, fp32, `gates` = (batch, 4*width)
 * pre-activations in PyTorch's order i, f, g, o;  c = f c_prev + i g,  h = o tanh(c).  The forward keeps the activated gates
 * (`act`, batch x 4*width) and tanh(c) for the backward, which turns the gradients with respect to h and c (either may be
 * null = zero) into those with respect to the pre-activations and c_prev.  Enqueued on `stream`; capturable in a HIP graph. */
int saa_lstm_cell_forward(int32_t device, int32_t batch, int32_t width, const float *gates_dev, const float *c_prev_dev,
                          float *h_dev, float *c_dev, float *act_dev, float *tanh_c_dev, void *stream);
int saa_lstm_cell_backward(int32_t device, int32_t batch, int32_t width, const float *act_dev, const float *tanh_c_dev,
                           const float *c_prev_dev, const float *dh_dev, const float *dc_next_dev, float *dgates_dev,
                           float *dc_prev_dev, void *stream);

/* A whole LSTM recurrence of the training pass in one launch, and its backward pass in one launch (`model_train`,
 * DNN_tools.py:103-165: encoder `nn.LSTM`, DNN_tools.py:32, and the decoder steps, :73-79; widths 50 = the encoder's hidden
 * size and 100 = the decoder's, Model_training.py:36-40 - other widths are refused and stay with PyTorch):
 *   gates_t = pre[b, t, :] + h_{t-1} W^T,  c_t = f c_{t-1} + i g,  h_t = o tanh(c_t)   over t = 0..steps-1 (reverse: downwards),
 * `pre_dev` (batch, steps, 4*width) = input projections + biases, `w_dev` (4*width, width) row-major, `h0_dev` / `c0_dev`
 * (batch, width) or null = zero.  The forward writes every h_t (`h_all_dev`, batch x steps x width; the last state is its
 * last processed row) and keeps c_t, the activated gates and tanh(c_t) for the backward, which turns the gradients with
 * respect to every h_t (`dh_all_dev`) and the final c (`dc_last_dev`, may be null) into those with respect to `pre`, h0 and
 * c0.  The weight gradient is one product outside: dW = dpre^T . h_prev over all rows and steps. */
int saa_lstm_recurrence_forward(int32_t device, int32_t batch, int32_t steps, int32_t width, int32_t reverse, const float *pre_dev,
                                const float *h0_dev, const float *c0_dev, const float *w_dev, float *h_all_dev, float *c_all_dev,
                                float *act_dev, float *tanh_c_dev, void *stream);
int saa_lstm_recurrence_backward(int32_t device, int32_t batch, int32_t steps, int32_t width, int32_t reverse,
                                 const float *dh_all_dev, const float *dc_last_dev, const float *c0_dev, const float *w_dev,
                                 const float *c_all_dev, const float *act_dev, const float *tanh_c_dev, float *dpre_dev,
                                 float *dh0_dev, float *dc0_dev, void *stream);

/* The three figures `model_train` / `model_test` accumulate per batch (DNN_tools.py:144-155,196-205) - the mean square error
 * of `out_dev` against `target_dev` (n fp32 elements each), 1 - mse / mean((y - mean y)^2) and 1 - mse / mean(y^2) - added to
 * the three doubles `sums3_dev`; sums in fp64.  `scratch3_dev`: three doubles, zero before the first call, left zero. */
int saa_train_stats(int32_t device, int64_t n, const float *out_dev, const float *target_dev, double *scratch3_dev,
                    double *sums3_dev, void *stream);

/* Timing aid for bench.py: runs `nsteps` saa_step steps bracketed by HIP events recorded on the
 * handle's stream and returns the elapsed milliseconds (kernel time incl. launch gaps). */
int saa_time_steps(saa_solver *s, int32_t nsteps, double *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* SAA_HIP_H */
