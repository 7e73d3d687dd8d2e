// C ABI of libsaa_hip.so (see include/saa_hip.h): handle management, host<->device marshalling in the
// caller's numbering, step sequencing.  All numerics live in saa_kernels.hip.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

#include "../../include/saa_hip.h"
#include "saa_device.h"
#include "saa_partition.h"
#include "saa_plan.h"
#include "saa_predictor.h"
#include "saa_setup.h"
#include "saa_topology.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(SAA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

constexpr int kLdsBudget = 160 * 1024;

// The few RCCL entry points the native exchange needs, resolved from the library already loaded in the
// process (nothing is linked: the build has no RCCL dependency).  Types as in rccl.h.
struct NcclUniqueId {
  char internal[128];
};
struct NcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclDouble = 8, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h

bool load_nccl(const char *path, NcclApi &api, std::string &err) {
  if (api.lib) return true;
  void *h = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    err = std::string("dlopen(") + (path ? path : "librccl.so") + "): " + dlerror();
    return false;
  }
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
    err = "RCCL library lacks ncclGetUniqueId/ncclCommInitRank/ncclAllReduce";
    return false;
  }
  api.lib = h;
  return true;
}
NcclApi g_nccl;

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
  }
  hipError_t upload(const std::vector<T> &h) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

}  // namespace

struct saa_solver {
  saa::Plan plan;
  int device = 0;
  int threads = 0, lds_bytes = 0;
  hipStream_t stream = nullptr;
  saa::DeviceMesh mesh{};
  saa::SharedMap shared{};
  saa::StepConsts consts{};
  int ramp = 1;
  double tn = 0.0;
  // device storage
  DevBuf<saa::BlockDesc> blocks;
  DevBuf<int32_t> halo_ids, tag, new_to_old, sh_node, sh_slot, sh_foreign, slot_sidx;
  DevBuf<uint64_t> conn;
  DevBuf<double> xyz, mass, fext, mass_node, fext_yz;
  DevBuf<double> dbuf[3];
  DevBuf<double> scratch[4];
  int i0 = 0, in_ = 1, i1 = 2;  // dbuf indices of d^n, d^(n-1), d^(n+1)
  double *iface = nullptr;
  bool pending = false;
  int32_t n_shared = 0, n_global_shared = 0;
  std::vector<double> host_tmp;
  void *comm = nullptr;  // ncclComm_t of the native exchange (saa_comm_init)
  int32_t comm_world = 0;
  // saa_step_synced as replayed HIP graphs: three steps (the rotation period of the state buffers) per graph, one graph
  // per rotation phase; the clock lives in device memory (tn3: the time of d^n of step j in slot j % 3)
  DevBuf<double> tn3;
  hipGraphExec_t sync_graph[3] = {nullptr, nullptr, nullptr};
  hipStream_t sync_graph_stream = nullptr;
  double *sync_graph_iface = nullptr;
  bool sync_graph_off = false;     // capture or instantiation failed once: eager launches for good
  bool sync_graph_wanted = true;   // saa_set_option("synced_graph")
  double wait_timeout_s = 30.0;    // saa_set_option("wait_timeout_s"): bound of every in-kernel wait for other workgroups / ranks
  // resident multi-step kernel (saa_device.h: PersistArgs)
  DevBuf<int32_t> ps_err;
  DevBuf<saa::PeerEntry> ps_entries;  // 2 x 3*n_nodes stamped displacements
  int32_t ps_lds = 0, ps_max_items = 0, ps_steps = 0;
  int32_t ps_lds_peer = 0;  // LDS of the PEER variant: the image plus the block's push / receive records (0: does not fit)
  bool ps_capable = false;  // plan fits LDS and all workgroups can be co-resident
  bool ps_capable_predict = false;  // ... and the PREDICT instantiation passed its own census (saa_step_predicted)
  bool ps_enabled = true;   // saa_set_resident_kernel
  hipEvent_t ps_event = nullptr;  // recorded after every resident launch while other handles share the device
  // trajectory recorder (saa_set_recorder)
  double *rec_traj = nullptr;
  int64_t rec_cols = 0, rec_index = 0;
  int32_t rec_every = 1;
  // deterministic mode (saa_set_deterministic)
  bool det = false;
  DevBuf<double> det_force;
  DevBuf<int64_t> det_off;
  DevBuf<int32_t> det_contrib;
  saa::DetLists detl{};
  double *ps_dbg = nullptr; // diagnostic builds (-DSAA_PERSIST_STAMPS): where the per-wave cycle counts go
  // direct peer exchange (saa_peer_export / saa_peer_attach)
  void *peer_mem = nullptr;          // this rank's exported allocation: flags + inbox (fine-grained)
  int32_t peer_world = 0;
  std::vector<void *> peer_open;     // mapped allocations of the neighbours
  DevBuf<int32_t> px_blk_off, px_node, px_sidx, px_nb_off, px_err;
  DevBuf<saa::PeerEntry *> px_dst;
  DevBuf<int64_t> px_pstride, px_recv;
  DevBuf<saa::PeerPushRec> px_push_rec;
  DevBuf<saa::PeerRecvRec> px_recv_rec;
  DevBuf<saa::PeerSecondRec> px_second_rec;
  DevBuf<saa::PeerMap> px_map;  // device copy of `peer` (the step kernel reads it from memory)
  DevBuf<unsigned long long> px_holders;
  DevBuf<double> px_own, px_test;
  std::vector<double> px_expected;   // self-test: expected sums
  saa::PeerMap peer{};
  bool peer_ready = false;
  unsigned peer_seq = 0;
  // Split stepping (plans of several rounds of workgroups, i.e. partitions beyond the resident kernel's capacity): the
  // blocks in three sets - left, right and the blocks between them - stepped on three streams, setup_split_stepping()
  DevBuf<saa::BlockDesc> split_blocks[3];
  int32_t split_n[3] = {0, 0, 0};
  hipStream_t split_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t split_ev[3] = {nullptr, nullptr, nullptr}, split_fork = nullptr;
  bool split_ok = false, split_wanted = true;

  void rotate() {
    const int old_n = in_;
    in_ = i0;
    i0 = i1;
    i1 = old_n;
    record_if_due();
  }
  // per-step paths: the state that has just become d^n is a column of the caller's trajectory if its step index is due
  void record_if_due() {
    if (rec_traj && rec_index % rec_every == 0 && rec_index / rec_every < rec_cols)
      saa::launch_record_column(plan.n_nodes, new_to_old.p, stream, dbuf[i0].p, rec_traj, rec_cols, rec_index / rec_every);
    ++rec_index;
  }
  void set_ramp() { consts.ramp = ramp ? (tn <= 1 ? tn : 1.0) : 1.0; }  // commons.py:7-11
  void release_all() {
    blocks.release(); halo_ids.release(); tag.release(); new_to_old.release(); sh_node.release();
    sh_slot.release(); sh_foreign.release(); slot_sidx.release(); conn.release(); xyz.release(); mass.release(); fext.release(); mass_node.release(); fext_yz.release();
    for (auto &b : dbuf) b.release();
    for (auto &b : scratch) b.release();
    ps_entries.release(); ps_err.release();
    det_force.release(); det_off.release(); det_contrib.release();
    for (auto &g : sync_graph) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
    tn3.release();
    for (void *q : peer_open) (void)hipIpcCloseMemHandle(q);
    peer_open.clear();
    if (peer_mem) (void)hipFree(peer_mem);
    peer_mem = nullptr;
    px_blk_off.release(); px_node.release(); px_sidx.release(); px_nb_off.release(); px_err.release();
    px_dst.release(); px_pstride.release(); px_recv.release(); px_push_rec.release(); px_recv_rec.release(); px_second_rec.release(); px_map.release(); px_holders.release(); px_own.release(); px_test.release();
    for (int j = 0; j < 3; ++j) {
      split_blocks[j].release();
      if (split_stream[j]) (void)hipStreamDestroy(split_stream[j]);
      if (split_ev[j]) (void)hipEventDestroy(split_ev[j]);
      split_stream[j] = nullptr;
      split_ev[j] = nullptr;
    }
    if (split_fork) (void)hipEventDestroy(split_fork);
    split_fork = nullptr;
    split_ok = false;
  }
};

namespace {

int pick_threads(const saa::Plan &plan, int requested) {
  if (requested > 0) return requested;
  int max_elem = 0;
  for (const auto &b : plan.blocks) max_elem = std::max(max_elem, 2 * b.n_elem);  // items ~ pairs
  // 1024 threads only when the whole grid is one wave of workgroups (one block per CU); several rounds of
  // 512-thread workgroups overlap better (8.2M tets: 84.6 us/step against 97.9)
  if (max_elem >= 4096 && plan.blocks.size() <= 256) return 1024;
  if (max_elem >= 2048) return 512;
  if (max_elem >= 512) return 256;
  if (max_elem >= 128) return 128;
  return 64;
}

int lds_bytes_of(const saa::Plan &plan) { return saa::lds_bytes_for(plan.max_local, plan.max_owned); }

void fill_stats(const saa::Plan &plan, int lds, int threads, saa_plan_stats *out) {
  out->n_blocks = static_cast<int32_t>(plan.blocks.size());
  out->max_owned = plan.max_owned;
  out->max_local = plan.max_local;
  out->n_elem_copies = plan.n_elem_copies;
  out->n_halo_total = plan.n_halo_total;
  out->lds_bytes = lds;
  out->threads = threads;
  out->lds_conflict_factor = plan.lds_conflict_factor;
  out->lds_atomic_conflict_factor = plan.lds_atomic_conflict_factor;
  out->n_items = plan.n_items;
  out->n_pairs = plan.n_pairs;
  out->n_by_construction = plan.n_by_construction;
  out->n_renumbered = plan.n_renumbered;
  out->reserved = 0;
}

bool build_fitting_plan(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                        int32_t block_nodes, saa::Plan &plan, std::string &err, const int32_t *extra_work = nullptr) {
  int32_t bn = block_nodes;  // <= 0: automatic (saa_plan.cpp)
  while (true) {
    if (!saa::build_plan(n_nodes, n_elems, xyz, tets, bn, plan, err, extra_work)) return false;
    if (lds_bytes_of(plan) <= kLdsBudget) return true;
    if (bn <= 0) bn = std::max(plan.max_owned, 16);
    if (bn <= 8) {
      err = "node blocks do not fit the 160 KiB LDS budget";
      return false;
    }
    bn /= 2;
  }
}

// host vector in caller order -> internal order
void permute_in(const saa::Plan &plan, const double *host, std::vector<double> &out) {
  out.resize(3 * static_cast<size_t>(plan.n_nodes));
  for (int32_t i = 0; i < plan.n_nodes; ++i) {
    const size_t o = 3 * static_cast<size_t>(plan.new_to_old[i]);
    out[3 * static_cast<size_t>(i) + 0] = host[o + 0];
    out[3 * static_cast<size_t>(i) + 1] = host[o + 1];
    out[3 * static_cast<size_t>(i) + 2] = host[o + 2];
  }
}

void permute_out(const saa::Plan &plan, const std::vector<double> &in, double *host) {
  for (int32_t i = 0; i < plan.n_nodes; ++i) {
    const size_t o = 3 * static_cast<size_t>(plan.new_to_old[i]);
    host[o + 0] = in[3 * static_cast<size_t>(i) + 0];
    host[o + 1] = in[3 * static_cast<size_t>(i) + 1];
    host[o + 2] = in[3 * static_cast<size_t>(i) + 2];
  }
}

int upload_permuted(saa_solver *s, const double *host, double *dev) {
  permute_in(s->plan, host, s->host_tmp);
  HIP_TRY(hipMemcpyAsync(dev, s->host_tmp.data(), s->host_tmp.size() * sizeof(double), hipMemcpyHostToDevice,
                         s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return SAA_OK;
}

int download_permuted(saa_solver *s, const double *dev, double *host) {
  s->host_tmp.resize(3 * static_cast<size_t>(s->plan.n_nodes));
  HIP_TRY(hipMemcpyAsync(s->host_tmp.data(), dev, s->host_tmp.size() * sizeof(double), hipMemcpyDeviceToHost,
                         s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  permute_out(s->plan, s->host_tmp, host);
  return SAA_OK;
}

// Per-node copy of the lumped mass if it is the same for the three dofs of every node, else none.
int refresh_nodal_mass(saa_solver *s, const std::vector<double> &mass_internal) {
  const int32_t n = s->plan.n_nodes;
  bool nodal = true;
  for (int32_t i = 0; i < n && nodal; ++i)
    nodal = mass_internal[3 * static_cast<size_t>(i)] == mass_internal[3 * static_cast<size_t>(i) + 1] &&
            mass_internal[3 * static_cast<size_t>(i)] == mass_internal[3 * static_cast<size_t>(i) + 2];
  s->mesh.mass_node = nullptr;
  if (!nodal) return SAA_OK;
  std::vector<double> mn(n);
  for (int32_t i = 0; i < n; ++i) mn[i] = mass_internal[3 * static_cast<size_t>(i)];
  if (!s->mass_node.p) HIP_TRY(s->mass_node.alloc(n));
  HIP_TRY(hipMemcpy(s->mass_node.p, mn.data(), n * sizeof(double), hipMemcpyHostToDevice));
  s->mesh.mass_node = s->mass_node.p;
  return SAA_OK;
}

// Per-node copy of the load if it has the form (0, v, v) on every node (exact test), else none.
int refresh_nodal_load(saa_solver *s, const std::vector<double> &f_internal) {
  const int32_t n = s->plan.n_nodes;
  bool yz = true;
  for (int32_t i = 0; i < n && yz; ++i)
    yz = f_internal[3 * static_cast<size_t>(i)] == 0.0 && !std::signbit(f_internal[3 * static_cast<size_t>(i)]) &&
         f_internal[3 * static_cast<size_t>(i) + 1] == f_internal[3 * static_cast<size_t>(i) + 2];
  s->mesh.fext_yz = nullptr;
  if (!yz) return SAA_OK;
  std::vector<double> v(n);
  for (int32_t i = 0; i < n; ++i) v[i] = f_internal[3 * static_cast<size_t>(i) + 1];
  if (!s->fext_yz.p) HIP_TRY(s->fext_yz.alloc(n));
  HIP_TRY(hipMemcpy(s->fext_yz.p, v.data(), n * sizeof(double), hipMemcpyHostToDevice));
  s->mesh.fext_yz = s->fext_yz.p;
  return SAA_OK;
}

int ensure_scratch(saa_solver *s, int count) {
  for (int i = 0; i < count; ++i)
    if (!s->scratch[i].p) HIP_TRY(s->scratch[i].alloc(3 * static_cast<size_t>(s->plan.n_nodes)));
  return SAA_OK;
}

constexpr int kPeerMaxWorld = 64;  // holder masks are 64 bits

// A wait inside the peer-exchange kernel timed out (a neighbour died or never attached).
int check_peer_error(saa_solver *s) {
  if (!s->peer_ready) return SAA_OK;
  int32_t e = 0;
  HIP_TRY(hipMemcpy(&e, s->px_err.p, sizeof(e), hipMemcpyDeviceToHost));
  if (e != 0) return fail(SAA_E_STATE, "peer exchange: timed out waiting for a neighbour rank's shared-node forces");
  return SAA_OK;
}

// Steps shorter than this use one launch per step (the resident kernel pays its set-up once per launch).
constexpr int32_t kPersistMinSteps = 8;

int check_persist_error(saa_solver *s) {
  if (!s->ps_capable) return SAA_OK;
  int32_t e = 0;
  HIP_TRY(hipMemcpy(&e, s->ps_err.p, sizeof(e), hipMemcpyDeviceToHost));
  if (e != 0) return fail(SAA_E_STATE, "resident step kernel: timed out waiting for a neighbouring workgroup");
  return SAA_OK;
}

// Resident launches of ONE device must not overlap: each needs every workgroup of its grid on the chip at once, and two
// of them (two handles or two streams in one process) would each hold a part of the CUs and wait for the rest until their
// bounded waits give up.  The cooperative-launch API used to serialise them per device; with plain launches this does:
// while more than one resident-capable handle lives on a device, every resident launch waits for the event recorded
// behind the previous one (when that came from another handle) and records its own.  One handle alone pays nothing.
constexpr int kMaxDevices = 64;
std::mutex g_resident_mutex;
struct ResidentDeviceState {
  int handles = 0;              // resident-capable handles alive on this device
  saa_solver *last = nullptr;   // whose launch was enqueued last (its ps_event marks the end of that launch)
} g_resident[kMaxDevices];

// Census: one launch of the very kernel (same variant, same registers, same LDS, same grid) in which every workgroup
// checks in and waits for all the others - the proof of co-residency that the stamped waits of the step loop rely on.
// 50 ms bound; a grid that does not fit keeps the one-launch-per-step kernel.
bool persistent_census(saa_solver *s, int lds, int mode) {
  const int32_t nb = static_cast<int32_t>(s->plan.blocks.size());
  DevBuf<int32_t> counter;
  bool ok = counter.upload(std::vector<int32_t>(1, 0)) == hipSuccess;
  saa::PersistArgs a{};
  a.census = counter.p;
  a.err = s->ps_err.p;
  a.timeout_ticks = 5000000;  // 50 ms of the 100 MHz wall clock
  a.max_items = s->ps_max_items;
  ok = ok && saa::launch_persistent_steps(s->mesh, s->threads, lds, s->stream, s->consts, a, mode) == hipSuccess;
  ok = ok && hipStreamSynchronize(s->stream) == hipSuccess;
  int32_t e = 1, seen = 0;
  ok = ok && hipMemcpy(&e, s->ps_err.p, sizeof(e), hipMemcpyDeviceToHost) == hipSuccess &&
       hipMemcpy(&seen, counter.p, sizeof(seen), hipMemcpyDeviceToHost) == hipSuccess;
  counter.release();
  if (!ok || e != 0 || seen != nb) {
    (void)hipGetLastError();
    const int32_t zero = 0;
    (void)hipMemcpy(s->ps_err.p, &zero, sizeof(zero), hipMemcpyHostToDevice);
    return false;
  }
  return true;
}

// Entry buffers and capacity check of the resident kernel; failure only disables it.
void setup_persistent(saa_solver *s) {
  s->ps_capable = false;
  if (const char *env = saa::diag_env("SAA_NO_PERSISTENT"))
    if (env[0] == '1') return;
  const saa::Plan &plan = s->plan;
  const int32_t nb = static_cast<int32_t>(plan.blocks.size());
  int32_t max_items = 1, max_halo = 1;
  for (const auto &b : plan.blocks) {
    max_items = std::max(max_items, b.n_elem);
    max_halo = std::max(max_halo, b.n_halo);
  }
  const int lds = saa::persistent_lds_bytes(plan.max_local, plan.max_owned, max_items, max_halo);
  if (lds == 0) return;
  // grid sizing: occupancy query clamped by the scalar-register rule (persistent_max_blocks), then the census below.
  // SAA_RESIDENT_TRUST_GRID=1 (diagnostic build, the census test) skips the first check so that an over-sized grid
  // reaches the census.
  const char *trust = saa::diag_env("SAA_RESIDENT_TRUST_GRID");
  const int max_blocks = saa::persistent_max_blocks(s->device, s->threads, lds);
  if (max_blocks <= 0 || (max_blocks < nb && !(trust && trust[0] == '1'))) return;
  const size_t n_entries = 2 * 3 * static_cast<size_t>(plan.n_nodes);
  if (s->ps_entries.alloc(n_entries) != hipSuccess || s->ps_err.upload(std::vector<int32_t>(1, 0)) != hipSuccess ||
      hipMemset(s->ps_entries.p, 0, n_entries * sizeof(saa::PeerEntry)) != hipSuccess) {  // stamp 0 = never written
    (void)hipGetLastError();
    s->ps_entries.release();
    return;
  }
  s->ps_lds = lds;
  s->ps_max_items = max_items;
  s->ps_steps = 0;
  // nothing else may occupy the CUs while the census counts (another handle's resident launch would make it fail)
  if (hipDeviceSynchronize() != hipSuccess || !persistent_census(s, lds, 0)) {
    (void)hipGetLastError();
    s->ps_entries.release();
    return;
  }
  s->ps_capable = true;
  // the PREDICT instantiation is another kernel (other register counts): it gets its own census, and only the predicted
  // resident path depends on it
  s->ps_capable_predict = persistent_census(s, lds, 1);
  if (s->device >= 0 && s->device < kMaxDevices && hipEventCreateWithFlags(&s->ps_event, hipEventDisableTiming) == hipSuccess) {
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    ++g_resident[s->device].handles;
  } else {
    (void)hipGetLastError();
    s->ps_event = nullptr;
    s->ps_capable = s->ps_capable_predict = false;  // without the ordering event the resident path stays off
    s->ps_entries.release();
  }
}

// The end of a handle's part in the per-device ordering (saa_destroy).
void retire_resident(saa_solver *s) {
  if (!s->ps_event) return;
  {
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    ResidentDeviceState &d = g_resident[s->device];
    if (d.last == s) d.last = nullptr;  // (saa_destroy has synchronised the stream: nothing of this handle is in flight)
    --d.handles;
  }
  (void)hipEventDestroy(s->ps_event);
  s->ps_event = nullptr;
}

// Exchange-free (or peer-exchange) steps through the resident kernel, in launches of at most
// kPersistChunk steps; *n_done = how many of the nsteps were taken that way (0 when the resident kernel does not
// apply, < nsteps if a launch was refused half-way: the caller takes the rest with one fused kernel per step).
constexpr int32_t kPersistChunk = 1000;

int try_persistent_steps(saa_solver *s, int32_t nsteps, const double *table_dev, int64_t table_row0, double *hist_dev,
                         int64_t hist_row0, int32_t *n_done, bool peer = false) {
  *n_done = 0;
  if (s->det || !s->ps_capable || !s->ps_enabled || nsteps < kPersistMinSteps || !s->mesh.mass_node || !s->mesh.fext_yz)
    return SAA_OK;
  if (peer && s->ps_lds_peer == 0) return SAA_OK;
  if (table_dev != nullptr && !s->ps_capable_predict) return SAA_OK;
  int32_t chunk = kPersistChunk;
  if (const char *env = saa::diag_env("SAA_PERSIST_CHUNK")) chunk = std::max(kPersistMinSteps, std::atoi(env));
  const double timeout_s = s->wait_timeout_s;
  while (*n_done < nsteps) {
    int32_t n = std::min(chunk, nsteps - *n_done);
    if (nsteps - *n_done - n > 0 && nsteps - *n_done - n < kPersistMinSteps) n = nsteps - *n_done;  // no tiny tail
    if (peer && s->peer_seq > 0xffffffffu - static_cast<uint32_t>(n) - 2u) return SAA_OK;  // wrap: per-step path
    if (static_cast<uint32_t>(s->ps_steps) > 0x7fff0000u) {  // stamps about to wrap: start over (0 = never written)
      HIP_TRY(hipMemsetAsync(s->ps_entries.p, 0, s->ps_entries.n * sizeof(saa::PeerEntry), s->stream));
      s->ps_steps = 0;
    }
    saa::PersistArgs a{};
    a.g0 = s->dbuf[s->i0].p;
    a.g1 = s->dbuf[s->in_].p;
    a.entries = s->ps_entries.p;
    a.entry_stride = 3 * static_cast<int64_t>(s->plan.n_nodes);
    a.step_base = s->ps_steps;
    a.nsteps = n;
    a.tn0 = s->tn;
    a.ramp_on = s->ramp;
    a.max_items = s->ps_max_items;
    a.table = table_dev;
    a.hist = hist_dev ? hist_dev : (table_dev == nullptr ? s->ps_dbg : nullptr);  // (ps_dbg: diagnostic builds only)
    a.table_row0 = table_row0 + *n_done;
    a.hist_row0 = hist_row0 + *n_done;
    a.width = 3 * static_cast<int64_t>(s->n_shared);
    a.err = s->ps_err.p;
    a.timeout_ticks = static_cast<int64_t>(timeout_s * 1e8);
    a.peer = peer ? s->px_map.p : nullptr;
    a.peer_rec_off = s->ps_lds;
    a.peer_seq_base = s->peer_seq;
    a.consts = s->consts;
    a.traj = s->rec_traj;
    a.new_to_old = s->new_to_old.p;
    a.traj_cols = s->rec_cols;
    a.step_index0 = s->rec_index;
    a.save_every = s->rec_every;
    hipError_t e = hipSuccess;
    {
      std::lock_guard<std::mutex> lock(g_resident_mutex);
      ResidentDeviceState &d = g_resident[s->device];
      const bool shared_device = d.handles > 1;
      if (shared_device && d.last != nullptr && d.last != s) e = hipStreamWaitEvent(s->stream, d.last->ps_event, 0);
      if (e == hipSuccess)
        e = saa::launch_persistent_steps(s->mesh, s->threads, peer ? s->ps_lds_peer : s->ps_lds, s->stream, s->consts, a,
                                         peer ? 2 : (table_dev != nullptr ? 1 : 0));
      if (e == hipSuccess && shared_device) {
        e = hipEventRecord(s->ps_event, s->stream);
        d.last = s;
      }
    }
    if (e != hipSuccess) {
      // a plain launch is only refused for reasons that will not go away (invalid configuration, a lost device): the
      // rest of this call and all later ones take one launch per step (co-residency itself is not checked here - the
      // census at set-up established it)
      (void)hipGetLastError();
      s->ps_capable = s->ps_capable_predict = false;
      return SAA_OK;
    }
    s->ps_steps = static_cast<int32_t>(static_cast<uint32_t>(s->ps_steps) + static_cast<uint32_t>(n));
    if (n & 1) std::swap(s->i0, s->in_);
    for (int32_t k = 0; k < n; ++k) s->tn = s->tn + s->consts.dt;  // the kernel advanced its copy the same way
    if (peer) s->peer_seq += static_cast<uint32_t>(n);
    s->rec_index += n;
    *n_done += n;
  }
  return SAA_OK;
}

// one step kernel (or pair of kernels in deterministic mode) from d^n, d^(n-1) into d^(n+1)
void launch_step(saa_solver *s, double *iface, const double *table_row, double *hist_row) {
  if (s->det)
    saa::launch_det_step(s->mesh, s->detl, s->threads, s->lds_bytes, s->stream, s->dbuf[s->i0].p, s->dbuf[s->in_].p,
                         s->dbuf[s->i1].p, iface, table_row, hist_row, s->consts, false);
  else
    saa::launch_fused_step(s->mesh, s->threads, s->lds_bytes, s->stream, s->dbuf[s->i0].p, s->dbuf[s->in_].p,
                           s->dbuf[s->i1].p, iface, table_row, hist_row, s->consts);
}

void launch_force(saa_solver *s, const double *d, double *f) {
  if (s->det)
    saa::launch_det_step(s->mesh, s->detl, s->threads, s->lds_bytes, s->stream, d, d, f, nullptr, nullptr, nullptr,
                         saa::StepConsts{}, true);
  else
    saa::launch_force_only(s->mesh, s->threads, s->lds_bytes, s->stream, d, f);
}

int check_launch();

// Split stepping.  A partition too large for the resident kernel runs one launch of the fused kernel per step, several
// rounds of workgroups each (8.2M tets on one GPU: 2048 blocks on 512 slots), and every launch boundary costs its tail -
// CUs idling while the last workgroups finish - and its ramp - all workgroups staging at once: eight steps' worth of
// blocks in ONE launch take 68.5 us per step against 75.9 (profiles/r04_fused_8Mtets_ablation.txt, variant 9).  A step
// only needs its NEIGHBOUR blocks' previous step, though.  So the blocks are split along the longest axis of the mesh into
// a left set L, a right set R and the set M of blocks whose halo reaches across the cut: L's halo nodes are owned by L or
// M, R's by R or M.  With X_s = "step s of the blocks of X":  L_s needs L_(s-1), M_(s-1);  R_s needs R_(s-1), M_(s-1);
// M_s needs all three of step s-1.  L, R and M run on a stream each, tied by events exactly so; L and R (half of the
// work each: the chain L_(s+1) <- M_s <- R_(s-1) makes the smaller of two unequal halves wait) fill each other's tails
// and ramps, because nothing but M ties their phases together.  Write-after-read on the three
// rotating state buffers: X_s overwrites d^(s-2) of its own nodes, last read as d^(n-1) by X_(s-1) (same stream) and as a
// halo value by step s-2 of a neighbour set, which the event chain has behind it.  Same kernel, same arithmetic per
// block: results do not depend on the split.
void setup_split_stepping(saa_solver *s, const double *xyz_caller) {
  s->split_ok = false;
  const saa::Plan &plan = s->plan;
  const int32_t nb = static_cast<int32_t>(plan.blocks.size());
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    return;
  }
  // Measured (tools/split_ab.py, 512-thread workgroups, 512 slots): 1024 blocks = two rounds lose 15 % (41.1 -> 47.5 us),
  // 1536 = three rounds are neutral (57.3), 2048 = four rounds gain 9.5 % (75.3 -> 68.1), 3328 = 6.5 rounds gain 4.4 %
  // (132.7 -> 126.9): from four rounds on.  (A resident-capable plan of that many blocks gets here only when the resident
  // kernel is switched off, saa_step.)
  if (nb < 8 * cus) return;
  // centroid of every block along the longest axis of the mesh
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int32_t i = 0; i < plan.n_nodes; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], xyz_caller[3 * static_cast<size_t>(i) + k]);
      hi[k] = std::max(hi[k], xyz_caller[3 * static_cast<size_t>(i) + k]);
    }
  int ax = 0;
  for (int k = 1; k < 3; ++k)
    if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
  std::vector<double> cen(nb, 0.0);
  std::vector<int32_t> start(nb);
  for (int32_t b = 0; b < nb; ++b) {
    const saa::BlockDesc &d = plan.blocks[b];
    start[b] = d.node_start;
    for (int32_t l = 0; l < d.n_owned; ++l)
      cen[b] += xyz_caller[3 * static_cast<size_t>(plan.new_to_old[d.node_start + l]) + ax];
    cen[b] /= std::max(d.n_owned, 1);
  }
  std::vector<int32_t> order(nb);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cen[a] < cen[b] || (cen[a] == cen[b] && a < b); });
  int64_t work = 0, acc = 0;
  for (const auto &d : plan.blocks) work += d.n_elem;
  std::vector<int8_t> side(nb, 1);  // 0 = left, 1 = right
  int left_pct = 50;  // (measured: 30 % no gain, 40 % 71.3, 45 % 69.8, 50 % 68.1 us per step at 8.2M tets against 75.4 unsplit)
  if (const char *env = saa::diag_env("SAA_SPLIT_LEFT_PCT")) left_pct = std::min(90, std::max(10, std::atoi(env)));
  for (int32_t b : order) {
    if (100 * acc >= static_cast<int64_t>(left_pct) * work) break;
    side[b] = 0;
    acc += plan.blocks[b].n_elem;
  }
  std::vector<int8_t> set(side);  // 0 = L, 2 = R (from side 1), 1 = M
  for (auto &v : set) v = v ? 2 : 0;
  for (int32_t b = 0; b < nb; ++b) {
    const saa::BlockDesc &d = plan.blocks[b];
    for (int32_t h = 0; h < d.n_halo; ++h) {
      const int32_t g = plan.halo_ids[d.halo_off + h];
      const int32_t owner = static_cast<int32_t>(std::upper_bound(start.begin(), start.end(), g) - start.begin()) - 1;
      if (side[owner] != side[b]) {
        set[b] = 1;
        break;
      }
    }
  }
  std::vector<saa::BlockDesc> lists[3];
  for (int32_t b = 0; b < nb; ++b) lists[set[b]].push_back(plan.blocks[b]);
  if (lists[0].size() < static_cast<size_t>(cus) || lists[2].size() < static_cast<size_t>(cus) || lists[1].empty()) return;
  for (int j = 0; j < 3; ++j) {
    s->split_n[j] = static_cast<int32_t>(lists[j].size());
    if (s->split_blocks[j].upload(lists[j]) != hipSuccess || hipStreamCreateWithFlags(&s->split_stream[j], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->split_ev[j], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return;  // (what was created is released with the handle)
    }
  }
  if (hipEventCreateWithFlags(&s->split_fork, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  s->split_ok = true;
}

// `nsteps` exchange-free steps of the fused kernel through the three streams; the caller's stream is forked at the start
// and joined at the end, so the call is ordered like any other on it.
int split_steps(saa_solver *s, int32_t nsteps) {
  HIP_TRY(hipEventRecord(s->split_fork, s->stream));
  for (int j = 0; j < 3; ++j) HIP_TRY(hipStreamWaitEvent(s->split_stream[j], s->split_fork, 0));
  saa::DeviceMesh m[3] = {s->mesh, s->mesh, s->mesh};
  for (int j = 0; j < 3; ++j) {
    m[j].blocks = s->split_blocks[j].p;
    m[j].n_blocks = s->split_n[j];
  }
  hipStream_t sl = s->split_stream[0], sm = s->split_stream[1], sr = s->split_stream[2];
  hipEvent_t el = s->split_ev[0], em = s->split_ev[1], er = s->split_ev[2];
  hipError_t err = hipSuccess;
  auto ok = [&](hipError_t e) {
    if (e != hipSuccess && err == hipSuccess) err = e;
    return err == hipSuccess;
  };
  for (int32_t k = 0; k < nsteps && err == hipSuccess; ++k) {
    s->set_ramp();
    const double *d0 = s->dbuf[s->i0].p, *dn = s->dbuf[s->in_].p;
    double *d1 = s->dbuf[s->i1].p;
    if (k > 0) {  // (the events still hold the records of step k - 1)
      if (!ok(hipStreamWaitEvent(sm, el, 0)) || !ok(hipStreamWaitEvent(sm, er, 0)) || !ok(hipStreamWaitEvent(sl, em, 0)) ||
          !ok(hipStreamWaitEvent(sr, em, 0)))
        break;
    }
    saa::launch_fused_step(m[1], s->threads, s->lds_bytes, sm, d0, dn, d1, nullptr, nullptr, nullptr, s->consts);
    saa::launch_fused_step(m[0], s->threads, s->lds_bytes, sl, d0, dn, d1, nullptr, nullptr, nullptr, s->consts);
    saa::launch_fused_step(m[2], s->threads, s->lds_bytes, sr, d0, dn, d1, nullptr, nullptr, nullptr, s->consts);
    if (!ok(hipEventRecord(em, sm)) || !ok(hipEventRecord(el, sl)) || !ok(hipEventRecord(er, sr))) break;
    s->rotate();
    s->tn = s->tn + s->consts.dt;  // Data_prepare.py:235
  }
  // the caller's stream continues behind all three sets - also when something failed on the way: nothing enqueued later
  // may overtake what is in flight
  for (int j = 0; j < 3; ++j) {
    (void)hipEventRecord(s->split_ev[j], s->split_stream[j]);
    (void)hipStreamWaitEvent(s->stream, s->split_ev[j], 0);
  }
  if (err != hipSuccess) return fail(SAA_E_HIP, std::string("split stepping: ") + hipGetErrorString(err));
  return check_launch();
}

// Synchronised steps (fused kernel -> ncclAllReduce -> finish kernel) as replayed HIP graphs: three launches per step
// and a collective's enqueue cost make the RCCL transport launch-bound from the host.  One graph holds THREE steps - after
// three the rotating state buffers are back where they were, so the captured pointers stay right - and there is one graph
// per rotation phase it can start from.  What changes from step to step inside a graph may not be a launch argument:
// the time of d^n (hence the ramp) lives in device memory, advanced by the finish kernel exactly as the host advances its
// copy.  Steps that record history rows or trajectory columns (their addresses change every step), deterministic mode and
// calls shorter than six steps keep the eager path.  *n_done: steps taken here (a multiple of three).
int try_synced_graphs(saa_solver *s, int32_t nsteps, bool wants_hist, int32_t *n_done) {
  *n_done = 0;
  if (s->sync_graph_off || wants_hist || s->rec_traj || s->det || nsteps < 6) return SAA_OK;
  if (!s->sync_graph_wanted) return SAA_OK;
  const size_t count = 3 * static_cast<size_t>(s->n_global_shared);
  if (s->sync_graph_stream != s->stream || s->sync_graph_iface != s->iface) {  // captured for another stream / buffer
    for (auto &g : s->sync_graph) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
    s->sync_graph_stream = s->stream;
    s->sync_graph_iface = s->iface;
  }
  if (!s->tn3.p && s->tn3.alloc(3) != hipSuccess) {
    (void)hipGetLastError();
    s->sync_graph_off = true;
    return SAA_OK;
  }
  const int phase = s->i0;  // which buffer holds d^n identifies the rotation phase
  if (!s->sync_graph[phase]) {
    hipGraph_t graph = nullptr;
    // relaxed mode: other threads of the process (PyTorch's collective watchdog polls events) must not break the capture
    hipError_t e = hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed);
    int nccl_rc = 0;
    if (e == hipSuccess) {
      int i0 = s->i0, in_ = s->in_, i1 = s->i1;
      for (int j = 0; j < 3; ++j) {
        const double *tn_j = s->ramp ? s->tn3.p + j : nullptr;
        saa::StepConsts k = s->consts;
        k.ramp = 1.0;  // (ramp off; with the ramp on the kernels overwrite it from the device clock)
        saa::launch_fused_step(s->mesh, s->threads, s->lds_bytes, s->stream, s->dbuf[i0].p, s->dbuf[in_].p, s->dbuf[i1].p,
                               s->iface, nullptr, nullptr, k, tn_j);
        if (count > 0 && nccl_rc == 0)
          nccl_rc = g_nccl.AllReduce(s->iface, s->iface, count, kNcclDouble, kNcclSum, s->comm, s->stream);
        saa::launch_iface_finish(s->mesh, s->shared, s->stream, s->dbuf[i0].p, s->dbuf[in_].p, s->dbuf[i1].p, s->iface,
                                 nullptr, k, s->tn3.p + j, s->tn3.p + (j + 1) % 3);
        const int old_n = in_;
        in_ = i0;
        i0 = i1;
        i1 = old_n;
      }
      e = hipStreamEndCapture(s->stream, &graph);
    }
    hipGraphExec_t exec = nullptr;
    if (e == hipSuccess && nccl_rc == 0 && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess || nccl_rc != 0 || !exec) {  // this RCCL / runtime cannot capture it: eager from now on
      (void)hipGetLastError();
      if (exec) (void)hipGraphExecDestroy(exec);
      s->sync_graph_off = true;
      return SAA_OK;
    }
    s->sync_graph[phase] = exec;
  }
  const int32_t replays = nsteps / 3;
  saa::launch_set_scalar(s->stream, s->tn3.p, s->tn);  // the device clock starts at this call's time
  for (int32_t r = 0; r < replays; ++r) {
    const hipError_t e = hipGraphLaunch(s->sync_graph[phase], s->stream);
    if (e != hipSuccess) return fail(SAA_E_HIP, std::string("saa_step_synced: hipGraphLaunch: ") + hipGetErrorString(e));
    for (int j = 0; j < 3; ++j) s->tn = s->tn + s->consts.dt;  // the finish kernels advanced their copy the same way
  }
  s->rec_index += 3 * static_cast<int64_t>(replays);
  *n_done = 3 * replays;  // (three rotations: i0 / in_ / i1 are where they were)
  return SAA_OK;
}

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SAA_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return SAA_OK;
}

}  // namespace

extern "C" {

const char *saa_last_error(void) { return g_last_error.c_str(); }

int32_t saa_abi_version(void) { return 10; }  // 4: saa_part_mesh_kway, saa_setup_fields; 5: saa_set_deterministic; 6: saa_device_copy_bandwidth; 7: saa_plan_stats grew; 8: saa_predictor_*, saa_topology_*; 9: saa_set_option, saa_plan_stats.n_renumbered; 10: saa_plan_host_check

int saa_device_copy_bandwidth(int32_t device, int64_t n_bytes, int32_t reps, double *bytes_per_s) {
  if (!bytes_per_s || n_bytes < 16 || reps < 1) return fail(SAA_E_ARG, "saa_device_copy_bandwidth: bad argument");
  HIP_TRY(hipSetDevice(device));
  const hipError_t e = saa::copy_bandwidth(device, n_bytes, reps, bytes_per_s);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(SAA_E_HIP, std::string("saa_device_copy_bandwidth: ") + hipGetErrorString(e));
  }
  return SAA_OK;
}

int saa_plan_host_stats(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets,
                        int32_t block_nodes, saa_plan_stats *out) {
  if (!out) return fail(SAA_E_ARG, "saa_plan_host_stats: null out");
  saa::Plan plan;
  std::string err;
  if (!build_fitting_plan(n_nodes, n_elems, xyz, tets, block_nodes, plan, err)) return fail(SAA_E_ARG, err);
  fill_stats(plan, lds_bytes_of(plan), pick_threads(plan, 0), out);
  return SAA_OK;
}

// Self-check of a block plan against the mesh it was built from (host only): what the step kernels rely on.
//   1. the internal numbering is a permutation of the caller's;
//   2. every work item names valid block-local nodes, its one or two tets are elements of the mesh (same four nodes) with the
//      orientation the mesh gives them (the sign of detJ is kept, Mat_construction.py:93), first-round items name owned nodes only;
//   3. every element appears exactly once in every block that owns one of its nodes, and nowhere else (owner computes: a node's
//      force is complete inside its block, nothing is summed across workgroups).
// *violations_out: how many of these checks failed (0 = the plan is what the kernels assume).
int saa_plan_host_check(int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets, int32_t block_nodes,
                        int64_t *violations_out) {
  if (!violations_out) return fail(SAA_E_ARG, "saa_plan_host_check: null out");
  saa::Plan plan;
  std::string err;
  if (!build_fitting_plan(n_nodes, n_elems, xyz, tets, block_nodes, plan, err)) return fail(SAA_E_ARG, err);
  int64_t bad = 0;
  {  // 1
    std::vector<char> seen(n_nodes, 0);
    for (int32_t i = 0; i < n_nodes; ++i) {
      const int32_t o = plan.new_to_old[i];
      if (o < 0 || o >= n_nodes || seen[o] || plan.old_to_new[o] != i) ++bad;
      else seen[o] = 1;
    }
  }
  const int32_t nb = static_cast<int32_t>(plan.blocks.size());
  std::vector<int32_t> start(nb);
  for (int32_t b = 0; b < nb; ++b) start[b] = plan.blocks[b].node_start;
  auto owner = [&](int32_t g) { return static_cast<int32_t>(std::upper_bound(start.begin(), start.end(), g) - start.begin()) - 1; };
  // elements by their sorted (caller-numbered) nodes
  struct Key {
    int32_t v[4];
    bool operator<(const Key &o) const { return std::lexicographical_compare(v, v + 4, o.v, o.v + 4); }
    bool operator==(const Key &o) const { return std::equal(v, v + 4, o.v); }
  };
  std::vector<std::pair<Key, int32_t>> elems(n_elems);
  for (int32_t e = 0; e < n_elems; ++e) {
    Key k;
    for (int a = 0; a < 4; ++a) k.v[a] = tets[4 * static_cast<size_t>(e) + a];
    std::sort(k.v, k.v + 4);
    elems[e] = {k, e};
  }
  std::sort(elems.begin(), elems.end());
  auto det = [&](const int32_t n[4]) {  // 6 x signed volume in the caller's numbering
    const double *p0 = xyz + 3 * static_cast<size_t>(n[0]), *p1 = xyz + 3 * static_cast<size_t>(n[1]),
                 *p2 = xyz + 3 * static_cast<size_t>(n[2]), *p3 = xyz + 3 * static_cast<size_t>(n[3]);
    const double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]},
                 c[3] = {p3[0] - p0[0], p3[1] - p0[1], p3[2] - p0[2]};
    return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
  };
  std::vector<std::pair<int32_t, int32_t>> copies;  // (element, block) of every tet found in the item lists
  for (int32_t b = 0; b < nb; ++b) {
    const saa::BlockDesc &d = plan.blocks[b];
    auto global = [&](uint16_t l) -> int32_t {  // caller's id of block-local node l, -1 if l is out of range
      if (l < d.n_owned) return plan.new_to_old[d.node_start + l];
      if (l < d.n_owned + d.n_halo) return plan.new_to_old[plan.halo_ids[d.halo_off + (l - d.n_owned)]];
      return -1;
    };
    for (int32_t i = 0; i < d.n_elem; ++i) {
      const uint16_t *it = &plan.conn[8 * static_cast<size_t>(d.elem_off + i)];
      if (it[5] == 2) continue;  // idle lane
      const int n_tets = it[5] == 1 ? 2 : 1;
      // A = (a; p, q, r), B = (b; p, r, q)
      const uint16_t loc[2][4] = {{it[0], it[1], it[2], it[3]}, {it[4], it[1], it[3], it[2]}};
      for (int t = 0; t < n_tets; ++t) {
        int32_t n[4];
        bool ok = true, any_owned = false, all_owned = true;
        for (int a = 0; a < 4; ++a) {
          n[a] = global(loc[t][a]);
          ok = ok && n[a] >= 0;
          any_owned = any_owned || loc[t][a] < d.n_owned;
          all_owned = all_owned && loc[t][a] < d.n_owned;
        }
        if (!ok || !any_owned || (i < d.n_interior && !all_owned)) {
          ++bad;
          continue;
        }
        Key k;
        std::copy(n, n + 4, k.v);
        std::sort(k.v, k.v + 4);
        auto hit = std::lower_bound(elems.begin(), elems.end(), std::make_pair(k, INT32_MIN));
        if (hit == elems.end() || !(hit->first == k)) {
          ++bad;  // not an element of the mesh
          continue;
        }
        const int32_t e = hit->second;
        int32_t orig[4];
        for (int a = 0; a < 4; ++a) orig[a] = tets[4 * static_cast<size_t>(e) + a];
        if ((det(n) > 0) != (det(orig) > 0)) ++bad;  // orientation changed
        copies.emplace_back(e, b);
      }
    }
  }
  std::sort(copies.begin(), copies.end());
  for (size_t i = 1; i < copies.size(); ++i) bad += copies[i] == copies[i - 1];  // an element twice in one block
  copies.erase(std::unique(copies.begin(), copies.end()), copies.end());
  std::vector<std::pair<int32_t, int32_t>> want;
  for (int32_t e = 0; e < n_elems; ++e) {
    int32_t bl[4];
    for (int a = 0; a < 4; ++a) bl[a] = owner(plan.old_to_new[tets[4 * static_cast<size_t>(e) + a]]);
    std::sort(bl, bl + 4);
    for (int a = 0; a < 4; ++a)
      if (a == 0 || bl[a] != bl[a - 1]) want.emplace_back(e, bl[a]);
  }
  std::vector<std::pair<int32_t, int32_t>> diff;
  std::set_symmetric_difference(copies.begin(), copies.end(), want.begin(), want.end(), std::back_inserter(diff));
  bad += static_cast<int64_t>(diff.size());
  if (static_cast<int64_t>(want.size()) != plan.n_elem_copies) ++bad;
  *violations_out = bad;
  return SAA_OK;
}

int saa_part_mesh_kway(int32_t n_parts, int32_t n_elems, int32_t n_nodes, const int32_t *tets, int32_t *epart_out,
                       saa_partition_stats *stats_out) {
  if (!epart_out && n_elems > 0) return fail(SAA_E_ARG, "saa_part_mesh_kway: null output");
  std::vector<int32_t> epart;
  saa::PartitionStats st;
  std::string err;
  if (!saa::partition_kway(n_parts, n_elems, n_nodes, tets, epart, st, err)) return fail(SAA_E_ARG, err);
  if (n_elems > 0) std::memcpy(epart_out, epart.data(), epart.size() * sizeof(int32_t));
  if (stats_out) {
    stats_out->face_cut = st.face_cut;
    stats_out->min_part = st.min_part;
    stats_out->max_part = st.max_part;
    stats_out->interface_nodes = st.interface_nodes;
  }
  return SAA_OK;
}

int saa_setup_fields(int32_t device, int32_t n_nodes, int32_t n_elems, const double *xyz, const int32_t *tets, double rho,
                     double fz, double *lumped_mass_out, double *f_pre_out, double *min_edge_out) {
  if (n_nodes <= 0 || n_elems < 0 || !xyz || (n_elems > 0 && !tets)) return fail(SAA_E_ARG, "saa_setup_fields: bad argument");
  if (4 * static_cast<int64_t>(n_elems) > INT32_MAX)  // the radix sort's pair count and the (element, corner) values are 32-bit
    return fail(SAA_E_CAPACITY, "saa_setup_fields: more than 2^29 elements in one call");
  for (int64_t i = 0; i < 4 * static_cast<int64_t>(n_elems); ++i)
    if (tets[i] < 0 || tets[i] >= n_nodes) return fail(SAA_E_ARG, "saa_setup_fields: node id out of range");
  HIP_TRY(hipSetDevice(device));
  const hipError_t e = saa::setup_fields(n_nodes, n_elems, xyz, tets, rho, fz, lumped_mass_out, f_pre_out, min_edge_out);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(SAA_E_HIP, std::string("saa_setup_fields: ") + hipGetErrorString(e));
  }
  return SAA_OK;
}

int saa_create(const saa_problem *pb, saa_solver **out) {
  if (!pb || !out) return fail(SAA_E_ARG, "saa_create: null argument");
  *out = nullptr;
  if (pb->n_nodes <= 0 || !pb->xyz || !pb->lumped_mass || !pb->f_ext || (pb->n_elems > 0 && !pb->tets))
    return fail(SAA_E_ARG, "saa_create: empty problem or null array");
  if ((pb->n_dirichlet > 0 && !pb->dirichlet_dofs) || (pb->n_shared > 0 && (!pb->shared_nodes || !pb->shared_slots)))
    return fail(SAA_E_ARG, "saa_create: null index list with non-zero length");
  if (pb->n_dirichlet < 0 || pb->n_shared < 0 || pb->n_global_shared < pb->n_shared)
    return fail(SAA_E_ARG, "saa_create: inconsistent list lengths");
  if (!(pb->dt > 0) || !std::isfinite(pb->dt) || !std::isfinite(pb->alpha) || !std::isfinite(pb->lambda_) ||
      !std::isfinite(pb->mu))
    return fail(SAA_E_ARG, "saa_create: dt must be positive and material parameters finite");
  if (pb->threads != 0 && (pb->threads % 64 != 0 || pb->threads < 64 || pb->threads > 1024))
    return fail(SAA_E_ARG, "saa_create: threads must be a multiple of 64 in [64,1024]");
  if (pb->n_global_shared >= (1 << (31 - saa::kTagSlotShift)))
    return fail(SAA_E_ARG, "saa_create: too many global shared nodes");
  for (int32_t i = 0; i < pb->n_dirichlet; ++i)
    if (pb->dirichlet_dofs[i] < 0 || pb->dirichlet_dofs[i] >= 3 * pb->n_nodes)
      return fail(SAA_E_ARG, "saa_create: Dirichlet dof out of range");
  for (int32_t i = 0; i < pb->n_shared; ++i)
    if (pb->shared_nodes[i] < 0 || pb->shared_nodes[i] >= pb->n_nodes || pb->shared_slots[i] < 0 ||
        pb->shared_slots[i] >= pb->n_global_shared)
      return fail(SAA_E_ARG, "saa_create: shared node or slot out of range");
  for (int64_t i = 0; i < 3 * static_cast<int64_t>(pb->n_nodes); ++i)
    if (!(pb->lumped_mass[i] != 0.0) || !std::isfinite(pb->lumped_mass[i]) || !std::isfinite(pb->f_ext[i]))
      return fail(SAA_E_ARG, "saa_create: lumped mass must be non-zero and loads finite");

  saa_solver *s = new (std::nothrow) saa_solver();
  if (!s) return fail(SAA_E_HIP, "saa_create: out of host memory");
  std::string err;
  // a shared node costs its workgroup a push to and a collect from the neighbour ranks on top of its elements
  // (tools/peer_loopback.py: ~1.9 us for ~95 shared nodes against ~11 us for ~5300 element copies): the blocks are
  // balanced with that in mind, so that interface blocks are not the ones every other block waits for
  std::vector<int32_t> extra;
  if (pb->n_shared > 0) {
    int32_t per_node = 16;  // 10 is the optimum with local latency (loop-back); xGMI's is longer
    if (const char *env = saa::diag_env("SAA_SHARED_NODE_WORK")) per_node = std::max(0, std::atoi(env));
    extra.assign(pb->n_nodes, 0);
    for (int32_t i = 0; i < pb->n_shared; ++i) extra[pb->shared_nodes[i]] = per_node;
  }
  if (!build_fitting_plan(pb->n_nodes, pb->n_elems, pb->xyz, pb->tets, pb->block_nodes, s->plan, err,
                          extra.empty() ? nullptr : extra.data())) {
    delete s;
    return fail(err.find("LDS") != std::string::npos ? SAA_E_CAPACITY : SAA_E_ARG, err);
  }
  // degenerate elements would divide by zero in the kernel (the reference would fail in np.linalg.inv)
  for (int32_t e = 0; e < pb->n_elems; ++e) {
    const double *x0 = pb->xyz + 3 * static_cast<int64_t>(pb->tets[4 * static_cast<int64_t>(e)]);
    double ed[3][3];
    for (int a = 0; a < 3; ++a) {
      const double *xa = pb->xyz + 3 * static_cast<int64_t>(pb->tets[4 * static_cast<int64_t>(e) + a + 1]);
      for (int c = 0; c < 3; ++c) ed[a][c] = xa[c] - x0[c];
    }
    const double det = ed[0][0] * (ed[1][1] * ed[2][2] - ed[1][2] * ed[2][1]) -
                       ed[0][1] * (ed[1][0] * ed[2][2] - ed[1][2] * ed[2][0]) +
                       ed[0][2] * (ed[1][0] * ed[2][1] - ed[1][1] * ed[2][0]);
    if (det == 0.0 || !std::isfinite(det)) {
      delete s;
      return fail(SAA_E_ARG, "saa_create: element " + std::to_string(e) + " is degenerate (detJ = 0)");
    }
  }

#define CREATE_TRY(expr)                                                                \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      s->release_all();                                                                 \
      delete s;                                                                         \
      return fail(SAA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    }                                                                                   \
  } while (0)

  s->device = pb->device;
  CREATE_TRY(hipSetDevice(pb->device));
  const saa::Plan &plan = s->plan;
  s->threads = pick_threads(plan, pb->threads);
  s->lds_bytes = lds_bytes_of(plan);
  CREATE_TRY(saa::configure_kernels(s->lds_bytes));

  const int32_t n = plan.n_nodes;
  std::vector<double> xyz(3 * static_cast<size_t>(n)), mass, fext;
  permute_in(plan, pb->xyz, xyz);
  permute_in(plan, pb->lumped_mass, mass);
  permute_in(plan, pb->f_ext, fext);
  std::vector<int32_t> tag(n, 0);
  for (int32_t i = 0; i < pb->n_dirichlet; ++i) {
    const int32_t dof = pb->dirichlet_dofs[i];
    tag[plan.old_to_new[dof / 3]] |= 1 << (dof % 3);
  }
  std::vector<int32_t> sh_node(pb->n_shared), sh_slot(pb->n_shared), foreign;
  std::vector<char> local_slot(pb->n_global_shared, 0);
  for (int32_t i = 0; i < pb->n_shared; ++i) {
    sh_node[i] = plan.old_to_new[pb->shared_nodes[i]];
    sh_slot[i] = pb->shared_slots[i];
    if (local_slot[sh_slot[i]] || (tag[sh_node[i]] & saa::kTagShared)) {
      s->release_all();
      delete s;
      return fail(SAA_E_ARG, "saa_create: duplicate shared node or interface slot");
    }
    local_slot[sh_slot[i]] = 1;
    tag[sh_node[i]] |= saa::kTagShared | (sh_slot[i] << saa::kTagSlotShift);
  }
  std::vector<int32_t> slot_sidx(pb->n_global_shared, -1);
  for (int32_t i = 0; i < pb->n_shared; ++i) slot_sidx[sh_slot[i]] = i;
  for (int32_t g = 0; g < pb->n_global_shared; ++g)
    if (!local_slot[g]) foreign.push_back(g);

  {
    // one entry of padding: the kernel's clamped prefetches read index 0 of possibly empty lists
    std::vector<int32_t> halo_padded(plan.halo_ids);
    halo_padded.push_back(0);
    std::vector<uint64_t> conn_padded(plan.conn_packed);
    conn_padded.push_back(0);
    CREATE_TRY(s->halo_ids.upload(halo_padded));
    CREATE_TRY(s->conn.upload(conn_padded));
  }
  CREATE_TRY(s->blocks.upload(plan.blocks));
  CREATE_TRY(s->new_to_old.upload(plan.new_to_old));
  CREATE_TRY(s->xyz.upload(xyz));
  CREATE_TRY(s->mass.upload(mass));
  CREATE_TRY(s->fext.upload(fext));
  CREATE_TRY(s->tag.upload(tag));
  CREATE_TRY(s->sh_node.upload(sh_node));
  CREATE_TRY(s->sh_slot.upload(sh_slot));
  CREATE_TRY(s->sh_foreign.upload(foreign));
  CREATE_TRY(s->slot_sidx.upload(slot_sidx));
  for (auto &b : s->dbuf) {
    CREATE_TRY(b.alloc(3 * static_cast<size_t>(n)));
    CREATE_TRY(hipMemset(b.p, 0, 3 * static_cast<size_t>(n) * sizeof(double)));
  }
#undef CREATE_TRY

  s->mesh.blocks = s->blocks.p;
  s->mesh.halo_ids = s->halo_ids.p;
  s->mesh.conn = reinterpret_cast<const uint2 *>(s->conn.p);
  s->mesh.xyz = s->xyz.p;
  s->mesh.mass = s->mass.p;
  s->mesh.fext = s->fext.p;
  s->mesh.tag = s->tag.p;
  if (refresh_nodal_mass(s, mass) != SAA_OK || refresh_nodal_load(s, fext) != SAA_OK) {
    s->release_all();
    delete s;
    return SAA_E_HIP;
  }
  s->mesh.slot_sidx = s->slot_sidx.p;
  s->mesh.lambda6 = pb->lambda_ / 6.0;
  s->mesh.mu6 = pb->mu / 6.0;
  s->mesh.n_blocks = static_cast<int32_t>(plan.blocks.size());
  s->mesh.n_nodes = n;
  s->mesh.max_local = plan.max_local;
  s->mesh.max_owned = plan.max_owned;
  s->shared.node = s->sh_node.p;
  s->shared.slot = s->sh_slot.p;
  s->shared.foreign_slot = s->sh_foreign.p;
  s->shared.n_shared = pb->n_shared;
  s->shared.n_foreign = static_cast<int32_t>(foreign.size());
  s->n_shared = pb->n_shared;
  s->n_global_shared = pb->n_global_shared;
  s->ramp = pb->ramp;
  // scalars exactly as Python forms them (Dynamic_solver.py:17): dt**2 is libm pow for numpy scalars
  s->consts.dt = pb->dt;
  s->consts.dt2 = std::pow(pb->dt, 2.0);
  s->consts.half_dt = pb->dt / 2;
  s->consts.alpha = pb->alpha;
  s->consts.half_alpha = 0.5 * pb->alpha;
  s->tn = 0.0;
  setup_persistent(s);
  setup_split_stepping(s, pb->xyz);
  *out = s;
  return SAA_OK;
}

int saa_destroy(saa_solver *s) {
  if (!s) return SAA_OK;
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);
  if (s->comm && g_nccl.CommDestroy) (void)g_nccl.CommDestroy(s->comm);
  retire_resident(s);
  s->release_all();
  delete s;
  return SAA_OK;
}

int saa_plan_stats_get(const saa_solver *s, saa_plan_stats *out) {
  if (!s || !out) return fail(SAA_E_ARG, "saa_plan_stats_get: null argument");
  fill_stats(s->plan, s->lds_bytes, s->threads, out);
  return SAA_OK;
}

int saa_set_stream(saa_solver *s, void *hip_stream) {
  if (!s) return fail(SAA_E_ARG, "saa_set_stream: null handle");
  s->stream = static_cast<hipStream_t>(hip_stream);
  return SAA_OK;
}

int saa_set_state(saa_solver *s, const double *d0_host, const double *dn_host, double tn) {
  if (!s || !d0_host || !dn_host) return fail(SAA_E_ARG, "saa_set_state: null argument");
  if (s->pending) return fail(SAA_E_STATE, "saa_set_state: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = upload_permuted(s, d0_host, s->dbuf[s->i0].p)) return rc;
  if (int rc = upload_permuted(s, dn_host, s->dbuf[s->in_].p)) return rc;
  s->tn = tn;
  return SAA_OK;
}

int saa_get_state(saa_solver *s, double *d0_host, double *dn_host, double *tn) {
  if (!s) return fail(SAA_E_ARG, "saa_get_state: null handle");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (int rc = check_persist_error(s)) return rc;
  if (int rc = check_peer_error(s)) return rc;
  if (d0_host)
    if (int rc = download_permuted(s, s->dbuf[s->i0].p, d0_host)) return rc;
  if (dn_host)
    if (int rc = download_permuted(s, s->dbuf[s->in_].p, dn_host)) return rc;
  if (tn) *tn = s->tn;
  return SAA_OK;
}

int saa_get_state_device(saa_solver *s, double *d0_dev, double *dn_dev) {
  if (!s) return fail(SAA_E_ARG, "saa_get_state_device: null handle");
  HIP_TRY(hipSetDevice(s->device));
  if (d0_dev) saa::launch_unpermute(s->plan.n_nodes, s->new_to_old.p, s->stream, s->dbuf[s->i0].p, d0_dev);
  if (dn_dev) saa::launch_unpermute(s->plan.n_nodes, s->new_to_old.p, s->stream, s->dbuf[s->in_].p, dn_dev);
  return check_launch();
}

int saa_set_loads(saa_solver *s, const double *f_ext_host, const double *lumped_mass_host) {
  if (!s) return fail(SAA_E_ARG, "saa_set_loads: null handle");
  HIP_TRY(hipSetDevice(s->device));
  if (lumped_mass_host) {
    for (int64_t i = 0; i < 3 * static_cast<int64_t>(s->plan.n_nodes); ++i)
      if (!(lumped_mass_host[i] != 0.0) || !std::isfinite(lumped_mass_host[i]))
        return fail(SAA_E_ARG, "saa_set_loads: lumped mass must be finite and non-zero");
    if (int rc = upload_permuted(s, lumped_mass_host, s->mass.p)) return rc;
    if (int rc = refresh_nodal_mass(s, s->host_tmp)) return rc;
  }
  if (f_ext_host) {
    if (int rc = upload_permuted(s, f_ext_host, s->fext.p)) return rc;
    if (int rc = refresh_nodal_load(s, s->host_tmp)) return rc;
  }
  return SAA_OK;
}

int saa_internal_force(saa_solver *s, const double *d_host, double *f_host) {
  if (!s || !d_host || !f_host) return fail(SAA_E_ARG, "saa_internal_force: null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = ensure_scratch(s, 2)) return rc;
  if (int rc = upload_permuted(s, d_host, s->scratch[0].p)) return rc;
  launch_force(s, s->scratch[0].p, s->scratch[1].p);
  if (int rc = check_launch()) return rc;
  return download_permuted(s, s->scratch[1].p, f_host);
}

int saa_internal_force_device(saa_solver *s, const double *d_dev, double *f_dev) {
  if (!s || !d_dev || !f_dev) return fail(SAA_E_ARG, "saa_internal_force_device: null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = ensure_scratch(s, 2)) return rc;
  saa::launch_permute(s->plan.n_nodes, s->new_to_old.p, s->stream, d_dev, s->scratch[0].p);
  launch_force(s, s->scratch[0].p, s->scratch[1].p);
  saa::launch_unpermute(s->plan.n_nodes, s->new_to_old.p, s->stream, s->scratch[1].p, f_dev);
  return check_launch();
}

int saa_cd_update(saa_solver *s, const double *f_int_host, const double *d0_host, const double *dn_host,
                  double tn, double *d1_host) {
  if (!s || !f_int_host || !d0_host || !dn_host || !d1_host) return fail(SAA_E_ARG, "saa_cd_update: null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (int rc = ensure_scratch(s, 4)) return rc;
  if (int rc = upload_permuted(s, f_int_host, s->scratch[0].p)) return rc;
  if (int rc = upload_permuted(s, d0_host, s->scratch[1].p)) return rc;
  if (int rc = upload_permuted(s, dn_host, s->scratch[2].p)) return rc;
  saa::StepConsts k = s->consts;
  k.ramp = s->ramp ? (tn <= 1 ? tn : 1.0) : 1.0;
  saa::launch_cd_update(s->mesh, s->stream, s->scratch[0].p, s->scratch[1].p, s->scratch[2].p, s->scratch[3].p, k);
  if (int rc = check_launch()) return rc;
  return download_permuted(s, s->scratch[3].p, d1_host);
}

int saa_step(saa_solver *s, int32_t nsteps) {
  if (!s || nsteps < 0) return fail(SAA_E_ARG, "saa_step: bad argument");
  if (s->pending) return fail(SAA_E_STATE, "saa_step: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  int32_t k0 = 0;
  if (int rc = try_persistent_steps(s, nsteps, nullptr, 0, nullptr, 0, &k0)) return rc;
  if (k0 == 0 && s->split_ok && s->split_wanted && !s->det && !s->rec_traj && nsteps >= 4) return split_steps(s, nsteps);
  for (int32_t k = k0; k < nsteps; ++k) {
    s->set_ramp();
    launch_step(s, nullptr, nullptr, nullptr);
    s->rotate();
    s->tn = s->tn + s->consts.dt;  // Data_prepare.py:235
  }
  return check_launch();
}

int saa_set_interface_buffer(saa_solver *s, double *iface_dev) {
  if (!s) return fail(SAA_E_ARG, "saa_set_interface_buffer: null handle");
  if (s->pending) return fail(SAA_E_STATE, "saa_set_interface_buffer: a synchronised step is in flight");
  s->iface = iface_dev;
  return SAA_OK;
}

int saa_step_begin(saa_solver *s) {
  if (!s) return fail(SAA_E_ARG, "saa_step_begin: null handle");
  if (s->pending) return fail(SAA_E_STATE, "saa_step_begin: previous step not finished");
  if (s->n_global_shared > 0 && !s->iface)
    return fail(SAA_E_STATE, "saa_step_begin: no interface buffer set");
  HIP_TRY(hipSetDevice(s->device));
  s->set_ramp();
  launch_step(s, s->iface, nullptr, nullptr);
  s->pending = true;
  return check_launch();
}

int saa_step_finish(saa_solver *s, double *hist_dev, int64_t hist_row) {
  if (!s) return fail(SAA_E_ARG, "saa_step_finish: null handle");
  if (!s->pending) return fail(SAA_E_STATE, "saa_step_finish: no step in flight");
  if (hist_dev && hist_row < 0) return fail(SAA_E_ARG, "saa_step_finish: negative history row");
  HIP_TRY(hipSetDevice(s->device));
  double *row = hist_dev ? hist_dev + hist_row * 3 * static_cast<int64_t>(s->n_shared) : nullptr;
  saa::launch_iface_finish(s->mesh, s->shared, s->stream, s->dbuf[s->i0].p, s->dbuf[s->in_].p, s->dbuf[s->i1].p,
                           s->iface, row, s->consts);
  s->pending = false;
  s->rotate();
  s->tn = s->tn + s->consts.dt;
  return check_launch();
}

int saa_comm_unique_id(const char *rccl_path, uint8_t id_out[128]) {
  if (!id_out) return fail(SAA_E_ARG, "saa_comm_unique_id: null output");
  std::string err;
  if (!load_nccl(rccl_path, g_nccl, err)) return fail(SAA_E_HIP, err);
  NcclUniqueId id;
  const int rc = g_nccl.GetUniqueId(&id);
  if (rc != 0) return fail(SAA_E_HIP, std::string("ncclGetUniqueId: ") + g_nccl.GetErrorString(rc));
  std::memcpy(id_out, id.internal, 128);
  return SAA_OK;
}

int saa_comm_init(saa_solver *s, const char *rccl_path, const uint8_t id_in[128], int32_t rank, int32_t world) {
  if (!s || !id_in || world < 1 || rank < 0 || rank >= world) return fail(SAA_E_ARG, "saa_comm_init: bad argument");
  if (s->comm) return fail(SAA_E_STATE, "saa_comm_init: communicator already initialised");
  if (s->n_global_shared > 0 && !s->iface) return fail(SAA_E_STATE, "saa_comm_init: set the interface buffer first");
  std::string err;
  if (!load_nccl(rccl_path, g_nccl, err)) return fail(SAA_E_HIP, err);
  HIP_TRY(hipSetDevice(s->device));
  NcclUniqueId id;
  std::memcpy(id.internal, id_in, 128);
  void *comm = nullptr;
  const int rc = g_nccl.CommInitRank(&comm, world, id, rank);
  if (rc != 0) return fail(SAA_E_HIP, std::string("ncclCommInitRank: ") + g_nccl.GetErrorString(rc));
  s->comm = comm;
  s->comm_world = world;
  return SAA_OK;
}

int saa_step_synced(saa_solver *s, int32_t nsteps, double *hist_dev, int64_t hist_row0) {
  if (!s || nsteps < 0 || (hist_dev && hist_row0 < 0)) return fail(SAA_E_ARG, "saa_step_synced: bad argument");
  if (!s->comm) return fail(SAA_E_STATE, "saa_step_synced: saa_comm_init has not been called");
  if (s->pending) return fail(SAA_E_STATE, "saa_step_synced: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  const size_t count = 3 * static_cast<size_t>(s->n_global_shared);
  const int64_t width = 3 * static_cast<int64_t>(s->n_shared);
  int32_t k0 = 0;
  if (int rc = try_synced_graphs(s, nsteps, hist_dev != nullptr, &k0)) return rc;
  for (int32_t k = k0; k < nsteps; ++k) {
    s->set_ramp();
    launch_step(s, s->iface, nullptr, nullptr);
    if (count > 0) {
      const int rc = g_nccl.AllReduce(s->iface, s->iface, count, kNcclDouble, kNcclSum, s->comm, s->stream);
      if (rc != 0) return fail(SAA_E_HIP, std::string("ncclAllReduce: ") + g_nccl.GetErrorString(rc));
    }
    saa::launch_iface_finish(s->mesh, s->shared, s->stream, s->dbuf[s->i0].p, s->dbuf[s->in_].p, s->dbuf[s->i1].p,
                             s->iface, hist_dev ? hist_dev + (hist_row0 + k) * width : nullptr, s->consts);
    s->rotate();
    s->tn = s->tn + s->consts.dt;
  }
  return check_launch();
}

int saa_peer_export(saa_solver *s, int32_t world, uint8_t handle_out[64], int32_t *order_out) {
  if (!s || !handle_out || (s->n_shared > 0 && !order_out) || world < 2 || world > kPeerMaxWorld)
    return fail(SAA_E_ARG, "saa_peer_export: bad argument (2 <= world <= 64)");
  if (s->peer_mem) return fail(SAA_E_STATE, "saa_peer_export: already exported");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = 2 * static_cast<size_t>(world) * 3 * std::max<size_t>(s->n_shared, 1) * sizeof(saa::PeerEntry);
  void *mem = nullptr;
  // fine-grained: stores of other agents become visible while kernels of this one are running
  HIP_TRY(hipExtMallocWithFlags(&mem, bytes, hipDeviceMallocFinegrained));
  hipError_t e = hipMemset(mem, 0, bytes);  // sequence number 0 is never sent
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, mem);
  if (e != hipSuccess) {
    (void)hipFree(mem);
    return fail(SAA_E_HIP, std::string("saa_peer_export: ") + hipGetErrorString(e));
  }
  // push order of this rank: position of every shared node in internal-node order
  {
    std::vector<int32_t> nodes(s->n_shared), idx(s->n_shared);
    if (s->n_shared > 0)
      e = hipMemcpy(nodes.data(), s->sh_node.p, nodes.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      (void)hipFree(mem);
      return fail(SAA_E_HIP, std::string("saa_peer_export: ") + hipGetErrorString(e));
    }
    for (int32_t i = 0; i < s->n_shared; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return nodes[a] < nodes[b]; });
    for (int32_t q = 0; q < s->n_shared; ++q) order_out[idx[q]] = q;
  }
  std::memcpy(handle_out, &h, 64);
  s->peer_mem = mem;
  s->peer_world = world;
  return SAA_OK;
}

// loopback (diagnostic, tools/peer_loopback.py): the "neighbours" live in this rank's own inbox, so that the cost of
// the push / collect phases of the PEER kernel can be timed on a one-GPU box (local instead of xGMI latency).
static int peer_attach_impl(saa_solver *s, int32_t rank, int32_t world, const uint8_t *handles, const int32_t *devices,
                            const int32_t *slot_counts, const int32_t *slots, const int32_t *orders, bool loopback) {
  if (!s || (!loopback && (!handles || !devices)) || !slot_counts || !slots || !orders || world < 2 || rank < 0 ||
      rank >= world)
    return fail(SAA_E_ARG, "saa_peer_attach: bad argument");
  if (!s->peer_mem || s->peer_world != world) return fail(SAA_E_STATE, "saa_peer_attach: saa_peer_export(world) first");
  if (s->peer_ready) return fail(SAA_E_STATE, "saa_peer_attach: already attached");
  if (s->pending) return fail(SAA_E_STATE, "saa_peer_attach: a synchronised step is in flight");
  if (slot_counts[rank] != s->n_shared) return fail(SAA_E_ARG, "saa_peer_attach: slot list of this rank has the wrong length");
  HIP_TRY(hipSetDevice(s->device));
  std::vector<int64_t> off(world + 1, 0);
  for (int p = 0; p < world; ++p) {
    if (slot_counts[p] < 0) return fail(SAA_E_ARG, "saa_peer_attach: negative slot count");
    off[p + 1] = off[p] + slot_counts[p];
  }
  const int32_t ngs = s->n_global_shared, nsh = s->n_shared;
  for (int64_t i = 0; i < off[world]; ++i)
    if (slots[i] < 0 || slots[i] >= ngs) return fail(SAA_E_ARG, "saa_peer_attach: slot out of range");
  const int32_t *mine = slots + off[rank];
  std::vector<int32_t> own_slots(nsh), own_nodes(nsh);
  if (nsh > 0) {
    HIP_TRY(hipMemcpy(own_slots.data(), s->sh_slot.p, nsh * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(own_nodes.data(), s->sh_node.p, nsh * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  for (int32_t i = 0; i < nsh; ++i)
    if (mine[i] != own_slots[i])
      return fail(SAA_E_ARG, "saa_peer_attach: slot list of this rank differs from saa_problem.shared_slots");
  // this rank's shared nodes sorted by internal node id: the nodes of one plan block are one range
  std::vector<int32_t> order(nsh);
  for (int32_t i = 0; i < nsh; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return own_nodes[a] < own_nodes[b]; });
  std::vector<int32_t> q_of_sidx(std::max<int32_t>(nsh, 1), 0), node(std::max<int32_t>(nsh, 1), 0), sidx(std::max<int32_t>(nsh, 1), 0);
  for (int32_t q = 0; q < nsh; ++q) {
    node[q] = own_nodes[order[q]];
    sidx[q] = order[q];
    q_of_sidx[order[q]] = q;
  }
  std::vector<int32_t> blk_off(s->plan.blocks.size() + 1, 0);
  {
    int32_t q = 0;
    for (size_t b = 0; b < s->plan.blocks.size(); ++b) {
      blk_off[b] = q;
      const int32_t end = s->plan.blocks[b].node_start + s->plan.blocks[b].n_owned;
      while (q < nsh && node[q] < end) ++q;
    }
    blk_off[s->plan.blocks.size()] = q;
    if (q != nsh) return fail(SAA_E_STATE, "saa_peer_attach: shared node outside every block");
  }
  std::vector<int32_t> slot_q(ngs, -1);  // interface slot -> q on this rank
  for (int32_t i = 0; i < nsh; ++i) slot_q[own_slots[i]] = q_of_sidx[i];
  std::vector<unsigned long long> holders(std::max<int32_t>(nsh, 1), 0ull);
  for (int32_t q = 0; q < nsh; ++q) holders[q] = 1ull << rank;
  struct Nb {
    int32_t p;
    saa::PeerEntry *dst;
    int64_t pstride, recv;
  };
  std::vector<std::vector<Nb>> nbs(std::max<int32_t>(nsh, 1));
  const int64_t per_me = 3 * static_cast<int64_t>(std::max<int32_t>(nsh, 1));
  for (int p = 0; p < world; ++p) {  // ascending: the neighbour entries of a node end up in rank order
    if (p == rank) continue;
    struct Common {
      int32_t q, order_p;
    };
    std::vector<Common> common;
    for (int32_t j = 0; j < slot_counts[p]; ++j) {
      const int32_t q = slot_q[slots[off[p] + j]];
      if (q >= 0) common.push_back({q, orders[off[p] + j]});
    }
    if (common.empty()) continue;
    void *base = s->peer_mem;
    if (!loopback) {
      hipIpcMemHandle_t h;
      std::memcpy(&h, handles + 64 * static_cast<size_t>(p), 64);
      std::string access = "same device";
      if (devices[p] != s->device) {
        int can = 0;
        const hipError_t ce = hipDeviceCanAccessPeer(&can, s->device, devices[p]);
        access = std::string("hipDeviceCanAccessPeer(") + std::to_string(s->device) + "," + std::to_string(devices[p]) +
                 ") = " + (ce == hipSuccess ? std::to_string(can) : std::string(hipGetErrorString(ce)));
        if (ce == hipSuccess && can) {
          const hipError_t pe = hipDeviceEnablePeerAccess(devices[p], 0);  // "already enabled" is fine
          access += std::string(", hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe);
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
            (void)hipGetLastError();
            return fail(SAA_E_HIP, "saa_peer_attach: rank " + std::to_string(p) + ": " + access);
          }
        } else if (ce != hipSuccess || !can) {
          (void)hipGetLastError();
          return fail(SAA_E_HIP, "saa_peer_attach: no peer access to the device of rank " + std::to_string(p) + ": " + access);
        }
        (void)hipGetLastError();
      }
      base = nullptr;
      const hipError_t oe = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
      if (oe != hipSuccess || !base) {
        (void)hipGetLastError();
        return fail(SAA_E_HIP, "saa_peer_attach: hipIpcOpenMemHandle(rank " + std::to_string(p) + "): " +
                                   hipGetErrorString(oe) + " (" + access + ")");
      }
      s->peer_open.push_back(base);
    }
    const int64_t per_p = 3 * static_cast<int64_t>(std::max<int32_t>(slot_counts[p], 1));
    saa::PeerEntry *inbox_p = static_cast<saa::PeerEntry *>(base);
    // position of every common node in MY push order (what I write) and in p's push order (what I read)
    std::vector<int32_t> t_send(common.size()), t_recv(common.size()), by(common.size());
    for (size_t i = 0; i < common.size(); ++i) by[i] = static_cast<int32_t>(i);
    std::sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return common[a].q < common[b].q; });
    for (size_t t = 0; t < by.size(); ++t) t_send[by[t]] = static_cast<int32_t>(t);
    std::sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return common[a].order_p < common[b].order_p; });
    for (size_t t = 0; t < by.size(); ++t) t_recv[by[t]] = static_cast<int32_t>(t);
    for (size_t i = 0; i < common.size(); ++i) {
      holders[common[i].q] |= 1ull << p;
      nbs[common[i].q].push_back({p, inbox_p + (loopback ? p : rank) * per_p + 3 * static_cast<int64_t>(t_send[i]), per_p * world,
                                  p * per_me + 3 * static_cast<int64_t>(t_recv[i])});
    }
  }
  std::vector<int32_t> nb_off(nsh + 1, 0);
  std::vector<saa::PeerEntry *> push_dst;
  std::vector<int64_t> push_pstride, recv_idx;
  for (int32_t q = 0; q < nsh; ++q) {
    nb_off[q] = static_cast<int32_t>(push_dst.size());
    for (const Nb &e : nbs[q]) {
      push_dst.push_back(e.dst);
      push_pstride.push_back(e.pstride);
      recv_idx.push_back(e.recv);
    }
  }
  nb_off[nsh] = static_cast<int32_t>(push_dst.size());
  std::vector<saa::PeerPushRec> push_rec(std::max<int32_t>(nsh, 1));
  std::vector<saa::PeerRecvRec> recv_rec(std::max<int32_t>(nsh, 1));
  std::vector<saa::PeerSecondRec> second_rec(std::max<int32_t>(nsh, 1), saa::PeerSecondRec{nullptr, 0, 0});
  for (size_t b = 0; b < s->plan.blocks.size(); ++b)
    for (int32_t q = blk_off[b]; q < blk_off[b + 1]; ++q) {
      if (nbs[q].size() >= 2) {
        const Nb &g = nbs[q][1];
        if (g.pstride > INT32_MAX || g.recv > INT32_MAX) return fail(SAA_E_CAPACITY, "saa_peer_attach: inbox too large");
        second_rec[q] = {g.dst, static_cast<int32_t>(g.pstride), static_cast<int32_t>(g.recv)};
      }
      const Nb f = nbs[q].empty() ? Nb{rank, nullptr, 0, 0} : nbs[q].front();
      if (f.pstride > INT32_MAX || f.recv > INT32_MAX) return fail(SAA_E_CAPACITY, "saa_peer_attach: inbox too large");
      const bool highest = (holders[q] >> rank) == 1ull;
      push_rec[q] = {f.dst, static_cast<int32_t>(f.pstride),
                     (node[q] - s->plan.blocks[b].node_start) | (static_cast<int32_t>(nbs[q].size()) << 16) |
                         (highest ? saa::kPeerInfoHighest : 0)};
      recv_rec[q] = {holders[q], static_cast<int32_t>(f.recv), sidx[q]};
    }
  if (push_dst.empty()) {  // keep the device arrays non-null
    push_dst.push_back(nullptr);
    push_pstride.push_back(0);
    recv_idx.push_back(0);
  }
  // self-test payload (node-sorted order) and its expected sums (caller's shared order): every holder contributes
  // rank + 1, the lowest-ranked holder adds 0.25*c so that the components differ
  std::vector<double> own(3 * static_cast<size_t>(std::max<int32_t>(nsh, 1)), 0.0);
  s->px_expected.assign(3 * static_cast<size_t>(nsh), 0.0);
  for (int32_t q = 0; q < nsh; ++q) {
    double sum = 0.0;
    for (int p = 0; p < world; ++p)
      if ((holders[q] >> p) & 1ull) sum += p + 1;
    const bool lowest = (holders[q] & ((1ull << rank) - 1ull)) == 0ull;
    for (int c = 0; c < 3; ++c) {
      own[3 * static_cast<size_t>(q) + c] = rank + 1 + (lowest ? 0.25 * c : 0.0);
      s->px_expected[3 * static_cast<size_t>(sidx[q]) + c] = sum + 0.25 * c;
    }
  }
  HIP_TRY(s->px_blk_off.upload(blk_off));
  HIP_TRY(s->px_node.upload(node));
  HIP_TRY(s->px_sidx.upload(sidx));
  HIP_TRY(s->px_nb_off.upload(nb_off));
  HIP_TRY(s->px_dst.upload(push_dst));
  HIP_TRY(s->px_pstride.upload(push_pstride));
  HIP_TRY(s->px_recv.upload(recv_idx));
  HIP_TRY(s->px_push_rec.upload(push_rec));
  HIP_TRY(s->px_recv_rec.upload(recv_rec));
  HIP_TRY(s->px_second_rec.upload(second_rec));
  HIP_TRY(s->px_holders.upload(holders));
  HIP_TRY(s->px_err.upload(std::vector<int32_t>(1, 0)));
  HIP_TRY(s->px_own.upload(own));
  HIP_TRY(s->px_test.upload(std::vector<double>(own.size(), 0.0)));
  saa::PeerMap &pm = s->peer;
  pm.blk_off = s->px_blk_off.p;
  pm.node = s->px_node.p;
  pm.sidx = s->px_sidx.p;
  pm.holders = s->px_holders.p;
  pm.push_rec = s->px_push_rec.p;
  pm.recv_rec = s->px_recv_rec.p;
  pm.second_rec = s->px_second_rec.p;
  pm.nb_off = s->px_nb_off.p;
  pm.push_dst = s->px_dst.p;
  pm.push_pstride = s->px_pstride.p;
  pm.recv_idx = s->px_recv.p;
  pm.inbox = static_cast<const saa::PeerEntry *>(s->peer_mem);
  pm.parity_stride = per_me * world;
  pm.err = s->px_err.p;
  pm.timeout_ticks = static_cast<int64_t>(s->wait_timeout_s * 1e8);  // wall_clock64(): 100 MHz
  pm.rank = rank;
  pm.world = world;
  pm.n_shared = nsh;
  HIP_TRY(s->px_map.upload(std::vector<saa::PeerMap>(1, pm)));
  s->peer_seq = 0;
  // resident PEER kernel: every block keeps its push / receive / second-neighbour records (16 bytes each per shared node)
  // behind its LDS image; the larger workgroup has to pass the census again
  s->ps_lds_peer = 0;
  if (s->ps_capable) {
    int32_t max_sh = 0;
    for (size_t b = 0; b + 1 < blk_off.size(); ++b) max_sh = std::max(max_sh, blk_off[b + 1] - blk_off[b]);
    const int lds = s->ps_lds + 48 * max_sh;
    if (lds <= 160 * 1024 && saa::configure_persistent_peer(lds) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
        persistent_census(s, lds, 2))
      s->ps_lds_peer = lds;
  }
  s->peer_ready = true;
  return SAA_OK;
}

int saa_peer_attach(saa_solver *s, int32_t rank, int32_t world, const uint8_t *handles, const int32_t *devices,
                    const int32_t *slot_counts, const int32_t *slots, const int32_t *orders) {
  return peer_attach_impl(s, rank, world, handles, devices, slot_counts, slots, orders, false);
}

int saa_peer_selftest(saa_solver *s, int32_t *ok) {
  if (!s || !ok) return fail(SAA_E_ARG, "saa_peer_selftest: null argument");
  *ok = 0;
  if (!s->peer_ready) return fail(SAA_E_STATE, "saa_peer_selftest: saa_peer_attach first");
  if (s->pending) return fail(SAA_E_STATE, "saa_peer_selftest: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  ++s->peer_seq;
  saa::launch_peer_selftest(s->peer, s->stream, s->px_own.p, s->px_test.p, s->peer_seq);
  if (int rc = check_launch()) return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  int32_t e = 0;
  HIP_TRY(hipMemcpy(&e, s->px_err.p, sizeof(e), hipMemcpyDeviceToHost));
  std::vector<double> got(3 * static_cast<size_t>(s->n_shared));
  if (!got.empty()) HIP_TRY(hipMemcpy(got.data(), s->px_test.p, got.size() * sizeof(double), hipMemcpyDeviceToHost));
  bool good = e == 0;
  for (size_t i = 0; i < got.size() && good; ++i) good = got[i] == s->px_expected[i];
  if (e != 0) {  // the caller falls back to the all-reduce: do not poison later saa_synchronize calls
    const int32_t zero = 0;
    HIP_TRY(hipMemcpy(s->px_err.p, &zero, sizeof(zero), hipMemcpyHostToDevice));
    s->peer_ready = false;
  }
  *ok = good ? 1 : 0;
  return SAA_OK;
}

int saa_step_peer(saa_solver *s, int32_t nsteps, double *hist_dev, int64_t hist_row0) {
  if (!s || nsteps < 0 || (hist_dev && hist_row0 < 0)) return fail(SAA_E_ARG, "saa_step_peer: bad argument");
  if (!s->peer_ready) return fail(SAA_E_STATE, "saa_step_peer: saa_peer_attach has not been called");
  if (s->det) return fail(SAA_E_STATE, "saa_step_peer: not available in deterministic mode (use saa_step_begin/finish)");
  if (s->pending) return fail(SAA_E_STATE, "saa_step_peer: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  const int64_t width = 3 * static_cast<int64_t>(s->n_shared);
  int32_t k0 = 0;
  if (int rc = try_persistent_steps(s, nsteps, nullptr, 0, hist_dev, hist_row0, &k0, true)) return rc;
  for (int32_t k = k0; k < nsteps; ++k) {
    s->set_ramp();
    if (++s->peer_seq == 0) ++s->peer_seq;  // 0 marks "never written"
    saa::launch_fused_step_peer(s->mesh, s->threads, s->lds_bytes, s->stream, s->dbuf[s->i0].p, s->dbuf[s->in_].p,
                                s->dbuf[s->i1].p, hist_dev ? hist_dev + (hist_row0 + k) * width : nullptr, s->consts,
                                s->px_map.p, s->peer_seq);
    s->rotate();
    s->tn = s->tn + s->consts.dt;
  }
  return check_launch();
}

int saa_step_predicted(saa_solver *s, int32_t nsteps, const double *table_dev, int64_t table_row0,
                       double *hist_dev, int64_t hist_row0) {
  if (!s || nsteps < 0 || table_row0 < 0 || (hist_dev && hist_row0 < 0))
    return fail(SAA_E_ARG, "saa_step_predicted: bad argument");
  if (s->n_shared > 0 && !table_dev) return fail(SAA_E_ARG, "saa_step_predicted: null table");
  if (s->pending) return fail(SAA_E_STATE, "saa_step_predicted: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  const int64_t width = 3 * static_cast<int64_t>(s->n_shared);
  int32_t k0 = 0;
  if (int rc = try_persistent_steps(s, nsteps, s->n_shared > 0 ? table_dev : nullptr, table_row0, hist_dev, hist_row0,
                                    &k0))
    return rc;
  for (int32_t k = k0; k < nsteps; ++k) {
    s->set_ramp();
    // halo overwrite + history record are fused into the step kernel's epilogue (one launch per step)
    launch_step(s, nullptr, s->n_shared > 0 ? table_dev + (table_row0 + k) * width : nullptr,
                hist_dev ? hist_dev + (hist_row0 + k) * width : nullptr);
    s->rotate();
    s->tn = s->tn + s->consts.dt;
  }
  return check_launch();
}

int saa_halo_gather(saa_solver *s, double *row_dev) {
  if (!s || (s->n_shared > 0 && !row_dev)) return fail(SAA_E_ARG, "saa_halo_gather: null argument");
  HIP_TRY(hipSetDevice(s->device));
  saa::launch_halo_gather(s->shared, s->stream, s->dbuf[s->i0].p, row_dev);
  return check_launch();
}

int saa_halo_scatter(saa_solver *s, const double *row_dev) {
  if (!s || (s->n_shared > 0 && !row_dev)) return fail(SAA_E_ARG, "saa_halo_scatter: null argument");
  HIP_TRY(hipSetDevice(s->device));
  saa::launch_halo_overwrite(s->shared, s->stream, row_dev, s->dbuf[s->i0].p, nullptr);
  return check_launch();
}

int saa_resident_kernel_info(const saa_solver *s, int32_t *capable, int32_t *lds_bytes, int32_t *steps_per_launch) {
  if (!s) return fail(SAA_E_ARG, "saa_resident_kernel_info: null handle");
  int32_t chunk = kPersistChunk;
  if (const char *env = saa::diag_env("SAA_PERSIST_CHUNK")) chunk = std::max(kPersistMinSteps, std::atoi(env));
  if (capable) *capable = s->ps_capable && s->ps_enabled && s->mesh.mass_node && s->mesh.fext_yz ? 1 : 0;
  if (lds_bytes) *lds_bytes = s->ps_lds;
  if (steps_per_launch) *steps_per_launch = chunk;
  return SAA_OK;
}

int saa_set_recorder(saa_solver *s, double *traj_dev, int64_t n_cols, int32_t save_every, int64_t next_step_index) {
  if (!s) return fail(SAA_E_ARG, "saa_set_recorder: null handle");
  if (traj_dev && (n_cols <= 0 || save_every <= 0 || next_step_index < 0))
    return fail(SAA_E_ARG, "saa_set_recorder: bad argument");
  if (s->pending) return fail(SAA_E_STATE, "saa_set_recorder: a synchronised step is in flight");
  s->rec_traj = traj_dev;
  s->rec_cols = traj_dev ? n_cols : 0;
  s->rec_every = traj_dev ? save_every : 1;
  s->rec_index = traj_dev ? next_step_index : 0;
  return SAA_OK;
}

int saa_set_deterministic(saa_solver *s, int32_t enable) {
  if (!s) return fail(SAA_E_ARG, "saa_set_deterministic: null handle");
  if (s->pending) return fail(SAA_E_STATE, "saa_set_deterministic: a synchronised step is in flight");
  HIP_TRY(hipSetDevice(s->device));
  if (enable && !s->det_off.p) {
    // per owned node (internal order): the item-force vectors addressed to it, ascending by item and slot
    const saa::Plan &plan = s->plan;
    std::vector<int64_t> off(static_cast<size_t>(plan.n_nodes) + 1, 0);
    auto each = [&](auto &&fn) {
      for (const saa::BlockDesc &b : plan.blocks)
        for (int32_t e = 0; e < b.n_elem; ++e) {
          const uint16_t *it = &plan.conn[8 * static_cast<size_t>(b.elem_off + e)];
          if (it[5] == 2) continue;  // null item
          const int slots = it[5] == 1 ? 5 : 4;
          for (int a = 0; a < slots; ++a)
            if (it[a] < b.n_owned) fn(b.node_start + it[a], (b.elem_off + e) * 8 + a);
        }
    };
    each([&](int32_t node, int32_t) { ++off[node + 1]; });
    for (int32_t i = 0; i < plan.n_nodes; ++i) off[i + 1] += off[i];
    if (plan.n_items >= (1ll << 28)) return fail(SAA_E_CAPACITY, "saa_set_deterministic: too many work items");
    std::vector<int32_t> contrib(static_cast<size_t>(off[plan.n_nodes]));
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    each([&](int32_t node, int32_t id) { contrib[cur[node]++] = id; });
    HIP_TRY(s->det_off.upload(off));
    HIP_TRY(s->det_contrib.upload(contrib.empty() ? std::vector<int32_t>(1, 0) : contrib));
    HIP_TRY(s->det_force.alloc(15 * static_cast<size_t>(std::max<int64_t>(plan.n_items, 1))));
    HIP_TRY(saa::configure_det_kernels(s->lds_bytes));
    s->detl.item_force = s->det_force.p;
    s->detl.contrib_off = s->det_off.p;
    s->detl.contrib = s->det_contrib.p;
  }
  s->det = enable != 0;
  return SAA_OK;
}

int saa_set_option(saa_solver *s, const char *name, double value) {
  if (!s || !name) return fail(SAA_E_ARG, "saa_set_option: null argument");
  const std::string key(name);
  if (key == "synced_graph") {
    s->sync_graph_wanted = value != 0.0;
    return SAA_OK;
  }
  if (key == "split_stepping") {
    s->split_wanted = value != 0.0;
    return SAA_OK;
  }
  if (key == "wait_timeout_s") {
    if (!(value > 0.0) || value > 3600.0) return fail(SAA_E_ARG, "saa_set_option: wait_timeout_s must be in (0, 3600]");
    s->wait_timeout_s = std::max(1e-7, value);
    if (s->px_map.p) {  // attached already: the step kernels read the bound from the device copy of the peer map
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipStreamSynchronize(s->stream));
      s->peer.timeout_ticks = static_cast<int64_t>(s->wait_timeout_s * 1e8);
      HIP_TRY(s->px_map.upload(std::vector<saa::PeerMap>(1, s->peer)));
    }
    return SAA_OK;
  }
  return fail(SAA_E_ARG, "saa_set_option: unknown option '" + key + "' (synced_graph, split_stepping, wait_timeout_s)");
}

int saa_set_resident_kernel(saa_solver *s, int32_t enable) {
  if (!s) return fail(SAA_E_ARG, "saa_set_resident_kernel: null handle");
  s->ps_enabled = enable != 0;
  return SAA_OK;
}

int saa_synchronize(saa_solver *s) {
  if (!s) return fail(SAA_E_ARG, "saa_synchronize: null handle");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (int rc = check_persist_error(s)) return rc;
  return check_peer_error(s);
}

int saa_time_steps(saa_solver *s, int32_t nsteps, double *elapsed_ms) {
  if (!s || !elapsed_ms || nsteps < 0) return fail(SAA_E_ARG, "saa_time_steps: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  HIP_TRY(hipEventRecord(a, s->stream));
  int rc = saa_step(s, nsteps);
  HIP_TRY(hipEventRecord(b, s->stream));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  *elapsed_ms = ms;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return rc;
}

// ---- partition bookkeeping of one rank (saa_topology.hip) ------------------------------------------------------
struct saa_topology {
  saa::RankTopology t;
};

int saa_topology_build(int32_t device, int32_t n_nodes, int32_t n_elems, const int32_t *tets, const int32_t *epart, int32_t rank,
                       int32_t n_parts, const double *xyz, int32_t n_facets, const int32_t *facets, double clamp_tol,
                       saa_topology **out) {
  if (!out) return fail(SAA_E_ARG, "saa_topology_build: null output");
  *out = nullptr;
  saa_topology *t = new (std::nothrow) saa_topology;
  if (!t) return fail(SAA_E_HIP, "saa_topology_build: out of host memory");
  std::string err;
  hipError_t e;
  try {
    e = saa::rank_topology(device, n_nodes, n_elems, tets, epart, rank, n_parts, xyz, n_facets, facets, clamp_tol, &t->t, err);
  } catch (const std::bad_alloc &) {
    delete t;
    return fail(SAA_E_HIP, "saa_topology_build: out of host memory");
  }
  if (e != hipSuccess) {
    delete t;
    return err.empty() ? fail(SAA_E_HIP, std::string("saa_topology_build: ") + hipGetErrorString(e)) : fail(SAA_E_ARG, err);
  }
  *out = t;
  return SAA_OK;
}

int saa_topology_sizes(const saa_topology *t, int32_t *sizes) {
  if (!t || !sizes) return fail(SAA_E_ARG, "saa_topology_sizes: null argument");
  const saa::RankTopology &r = t->t;
  const size_t n[6] = {r.elements.size(), r.nodes.size(), r.shared_nodes.size(), r.global_shared.size(),
                       r.dirichlet_nodes.size(), r.dirichlet_local.size()};
  for (int i = 0; i < 6; ++i) sizes[i] = static_cast<int32_t>(n[i]);
  return SAA_OK;
}

int saa_topology_get(const saa_topology *t, int32_t *elements, int32_t *nodes, int32_t *cells_local, int32_t *shared_nodes,
                     int32_t *shared_local, int32_t *shared_slots, int32_t *global_shared, int32_t *dirichlet_nodes,
                     int32_t *dirichlet_local) {
  if (!t) return fail(SAA_E_ARG, "saa_topology_get: null handle");
  const saa::RankTopology &r = t->t;
  const std::pair<const std::vector<int32_t> *, int32_t *> io[9] = {
      {&r.elements, elements},         {&r.nodes, nodes},
      {&r.cells_local, cells_local},   {&r.shared_nodes, shared_nodes},
      {&r.shared_local, shared_local}, {&r.shared_slots, shared_slots},
      {&r.global_shared, global_shared}, {&r.dirichlet_nodes, dirichlet_nodes},
      {&r.dirichlet_local, dirichlet_local}};
  for (const auto &p : io)
    if (p.second && !p.first->empty()) std::memcpy(p.second, p.first->data(), p.first->size() * sizeof(int32_t));
  return SAA_OK;
}

int saa_topology_destroy(saa_topology *t) {
  delete t;
  return SAA_OK;
}

// ---- shared-node predictor (saa_predictor.hip) ----------------------------------------------------------------
struct saa_predictor {
  saa::Predictor *impl = nullptr;
};

int saa_predictor_create(int32_t device, int32_t input_size, int32_t hidden_size, int32_t n_past, int32_t n_future,
                         int32_t filter_size, const float *const *weights, int32_t n_weights, saa_predictor **out) {
  if (!out) return fail(SAA_E_ARG, "saa_predictor_create: null output");
  *out = nullptr;
  if (n_weights != saa::kPredictorWeights)
    return fail(SAA_E_ARG, "saa_predictor_create: expected the 22 tensors of the reference's state_dict");
  std::string err;
  saa::Predictor *impl = nullptr;
  const saa::PredictorShape sh{input_size, hidden_size, n_past, n_future, filter_size};
  const hipError_t e = saa::predictor_create(device, sh, weights, &impl, err);
  if (e != hipSuccess)
    return err.empty() ? fail(SAA_E_HIP, std::string("saa_predictor_create: ") + hipGetErrorString(e)) : fail(SAA_E_ARG, err);
  saa_predictor *p = new (std::nothrow) saa_predictor;
  if (!p) {
    saa::predictor_destroy(impl);
    return fail(SAA_E_HIP, "saa_predictor_create: out of host memory");
  }
  p->impl = impl;
  *out = p;
  return SAA_OK;
}

int saa_predictor_predict(saa_predictor *p, const double *hist_dev, int64_t hist_rows, int64_t ld_hist, int64_t n,
                          double scale_max, double scale_min, double *table_dev, int64_t ld_table, void *stream) {
  if (!p || !p->impl || !hist_dev || !table_dev) return fail(SAA_E_ARG, "saa_predictor_predict: null argument");
  const saa::PredictorShape &sh = saa::predictor_shape(p->impl);
  const int64_t window = (int64_t)sh.n_past * sh.filter;
  if (ld_hist < sh.input_size || ld_table < sh.input_size)
    return fail(SAA_E_ARG, "saa_predictor_predict: row stride below input_size");
  // (the GEMM kernel forms a tile's row offsets in 32 bits: 63 rows of ld_hist doubles + one row's columns)
  if (63 * ld_hist * 8 + (int64_t)sh.input_size * 8 >= (1ll << 31))
    return fail(SAA_E_ARG, "saa_predictor_predict: history row stride too large (63 rows of it must stay below 2 GiB)");
  if (n < window || n > hist_rows)
    return fail(SAA_E_ARG, "saa_predictor_predict: rows [n - n_past*filter, n) are not inside the history");
  if (!(scale_max - scale_min != 0.0)) return fail(SAA_E_ARG, "saa_predictor_predict: scale_max == scale_min");
  HIP_TRY(hipSetDevice(saa::predictor_device(p->impl)));
  HIP_TRY(saa::predictor_predict(p->impl, hist_dev, ld_hist, n, scale_max, scale_min, table_dev, ld_table,
                                 static_cast<hipStream_t>(stream)));
  return SAA_OK;
}

int saa_predictor_destroy(saa_predictor *p) {
  if (!p) return SAA_OK;
  saa::predictor_destroy(p->impl);
  delete p;
  return SAA_OK;
}

int saa_lstm_cell_forward(int32_t device, int32_t batch, int32_t width, const float *gates_dev, const float *c_prev_dev,
                          float *h_dev, float *c_dev, float *act_dev, float *tanh_c_dev, void *stream) {
  if (batch < 0 || width < 0 || (batch > 0 && width > 0 && (!gates_dev || !c_prev_dev || !h_dev || !c_dev || !act_dev || !tanh_c_dev)))
    return fail(SAA_E_ARG, "saa_lstm_cell_forward: bad argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(saa::lstm_cell_forward(batch, width, gates_dev, c_prev_dev, h_dev, c_dev, act_dev, tanh_c_dev,
                                 static_cast<hipStream_t>(stream)));
  return SAA_OK;
}

int saa_lstm_cell_backward(int32_t device, int32_t batch, int32_t width, const float *act_dev, const float *tanh_c_dev,
                           const float *c_prev_dev, const float *dh_dev, const float *dc_next_dev, float *dgates_dev,
                           float *dc_prev_dev, void *stream) {
  if (batch < 0 || width < 0 || (batch > 0 && width > 0 && (!act_dev || !tanh_c_dev || !c_prev_dev || !dgates_dev || !dc_prev_dev)))
    return fail(SAA_E_ARG, "saa_lstm_cell_backward: bad argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(saa::lstm_cell_backward(batch, width, act_dev, tanh_c_dev, c_prev_dev, dh_dev, dc_next_dev, dgates_dev, dc_prev_dev,
                                  static_cast<hipStream_t>(stream)));
  return SAA_OK;
}

int saa_lstm_recurrence_forward(int32_t device, int32_t batch, int32_t steps, int32_t width, int32_t reverse, const float *pre_dev,
                                const float *h0_dev, const float *c0_dev, const float *w_dev, float *h_all_dev, float *c_all_dev,
                                float *act_dev, float *tanh_c_dev, void *stream) {
  if (batch < 0 || steps < 0 || (batch > 0 && steps > 0 && (!pre_dev || !w_dev || !h_all_dev || !c_all_dev || !act_dev || !tanh_c_dev)))
    return fail(SAA_E_ARG, "saa_lstm_recurrence_forward: bad argument");
  if (width != 50 && width != 100) return fail(SAA_E_ARG, "saa_lstm_recurrence_forward: width must be 50 or 100");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(saa::lstm_rec_forward(batch, steps, width, reverse, pre_dev, h0_dev, c0_dev, w_dev, h_all_dev, c_all_dev, act_dev,
                                tanh_c_dev, static_cast<hipStream_t>(stream)));
  return SAA_OK;
}

int saa_lstm_recurrence_backward(int32_t device, int32_t batch, int32_t steps, int32_t width, int32_t reverse,
                                 const float *dh_all_dev, const float *dc_last_dev, const float *c0_dev, const float *w_dev,
                                 const float *c_all_dev, const float *act_dev, const float *tanh_c_dev, float *dpre_dev,
                                 float *dh0_dev, float *dc0_dev, void *stream) {
  if (batch < 0 || steps < 0 ||
      (batch > 0 && steps > 0 && (!dh_all_dev || !w_dev || !c_all_dev || !act_dev || !tanh_c_dev || !dpre_dev || !dh0_dev || !dc0_dev)))
    return fail(SAA_E_ARG, "saa_lstm_recurrence_backward: bad argument");
  if (width != 50 && width != 100) return fail(SAA_E_ARG, "saa_lstm_recurrence_backward: width must be 50 or 100");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(saa::lstm_rec_backward(batch, steps, width, reverse, dh_all_dev, dc_last_dev, c0_dev, w_dev, c_all_dev, act_dev,
                                 tanh_c_dev, dpre_dev, dh0_dev, dc0_dev, static_cast<hipStream_t>(stream)));
  return SAA_OK;
}

int saa_train_stats(int32_t device, int64_t n, const float *out_dev, const float *target_dev, double *scratch3_dev,
                    double *sums3_dev, void *stream) {
  if (n < 0 || (n > 0 && (!out_dev || !target_dev || !scratch3_dev || !sums3_dev)))
    return fail(SAA_E_ARG, "saa_train_stats: bad argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(saa::train_stats(n, out_dev, target_dev, scratch3_dev, sums3_dev, static_cast<hipStream_t>(stream)));
  return SAA_OK;
}

#ifdef SAA_DIAGNOSTICS
// Diagnostic build only (libsaa_hip_diag.so, -DSAA_DIAGNOSTICS; never in the product library): time `nsteps`
// launches of an ablated step kernel (state is not rotated; outputs are meaningless).  Used by tools/ablate.py only.
int saa_debug_time_ablated(saa_solver *s, int32_t variant, int32_t nsteps, double *elapsed_ms) {
  if (!s || !elapsed_ms) return fail(SAA_E_ARG, "saa_debug_time_ablated: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  // variant 8 writes 12 stamps per wave into scratch[0] (needs 12*8*waves bytes <= 24*n_nodes: checked)
  if (int rc = ensure_scratch(s, 1)) return rc;
  if (variant == 8 && 12ull * s->mesh.n_blocks * (s->threads / 64) > 3ull * s->plan.n_nodes)
    return fail(SAA_E_ARG, "saa_debug_time_ablated: scratch too small for stamps");
  s->set_ramp();
  HIP_TRY(hipEventRecord(a, s->stream));
  for (int32_t k = 0; k < nsteps; ++k)
    saa::launch_fused_step_ablated(variant, s->mesh, s->threads, s->lds_bytes, s->stream, s->dbuf[s->i0].p,
                                   s->dbuf[s->in_].p, s->dbuf[s->i1].p, s->consts, s->scratch[0].p);
  HIP_TRY(hipEventRecord(b, s->stream));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  *elapsed_ms = ms;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return check_launch();
}

#endif  // SAA_DIAGNOSTICS

int saa_peer_attach_loopback(saa_solver *s, int32_t world) {
  if (!s || world < 2 || world > 8 || s->n_shared <= 0) return fail(SAA_E_ARG, "saa_peer_attach_loopback: bad argument");
  uint8_t handle[64];
  std::vector<int32_t> order(s->n_shared);
  if (int rc = saa_peer_export(s, world, handle, order.data())) return rc;
  std::vector<int32_t> my_slots(s->n_shared), counts(world, s->n_shared), slots, orders;
  HIP_TRY(hipMemcpy(my_slots.data(), s->sh_slot.p, my_slots.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int p = 0; p < world; ++p) {
    slots.insert(slots.end(), my_slots.begin(), my_slots.end());
    orders.insert(orders.end(), order.begin(), order.end());
  }
  return peer_attach_impl(s, 0, world, nullptr, nullptr, counts.data(), slots.data(), orders.data(), true);
}

#ifdef SAA_DIAGNOSTICS
int saa_debug_time_peer(saa_solver *s, int32_t nsteps, double *elapsed_ms) {
  if (!s || !elapsed_ms || nsteps < 0) return fail(SAA_E_ARG, "saa_debug_time_peer: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  HIP_TRY(hipEventRecord(a, s->stream));
  int rc = saa_step_peer(s, nsteps, nullptr, 0);
  HIP_TRY(hipEventRecord(b, s->stream));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  *elapsed_ms = ms;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return rc ? rc : check_peer_error(s);
}

// Diagnostic (tools/persist_stamps.py): buffer of 8 * waves uint64 that a -DSAA_PERSIST_STAMPS build of the resident
// kernel fills with per-wave cycle counts during plain saa_step calls; the product build never touches it.
int saa_debug_set_stamp_buffer(saa_solver *s, double *buf_dev) {
  if (!s) return fail(SAA_E_ARG, "saa_debug_set_stamp_buffer: null handle");
  s->ps_dbg = buf_dev;
  return SAA_OK;
}

// Diagnostic: copies the stamps of the last variant-8 launch to the host (n = 12 * blocks * waves values).
int saa_debug_read_stamps(saa_solver *s, unsigned long long *out, int64_t n) {
  if (!s || !out || !s->scratch[0].p) return fail(SAA_E_ARG, "saa_debug_read_stamps: bad argument");
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(out, s->scratch[0].p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SAA_OK;
}
#endif  // SAA_DIAGNOSTICS

}  // extern "C"
