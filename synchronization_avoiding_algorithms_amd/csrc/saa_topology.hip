// Partition bookkeeping of one rank as O(N) device passes plus radix sorts (shared-node
// discovery and Dirichlet detection of Data_prepare.py:104-144 as device kernels).  The reference's lists are ORDERED -
// nodes in the order a sweep over the rank's elements first meets them (rankwise_dist, Distributed_tools.py:14-24), shared
// nodes in the order a sweep over the other ranks' lists, rank by rank, first meets them (find_shared_nodes, :29-40) - and
// the local numbering, hence every array of the solver, follows from those orders.  An order "by first occurrence in a
// sweep" is a sort by the position of the first occurrence, and that position is a minimum over occurrences:
//   key1[v] = min over (element e of my part, corner c) with tets[e][c] == v  of  4 e + c       -> my nodes, first-touch order
//   q*[v]   = the lowest other part that holds v (from a bit mask of holders per node)
//   key2[v] = min over (element e of part q*[v], corner c) with tets[e][c] == v  of  4 e + c
//   my shared nodes sorted by (q*, key2)                                                        -> find_shared_nodes' order
// (v's first occurrence in part q's node list precedes w's iff its first-touch key in q's sweep is smaller).
// Minima by integer atomicMin, orders by hipCUB radix sorts of (key, node) pairs with absent nodes keyed to the maximum.
#include "saa_topology.h"

#include <hipcub/hipcub.hpp>

namespace saa {

namespace {

constexpr uint32_t kNone = 0xffffffffu;
constexpr unsigned long long kNone64 = ~0ull;

__global__ void mark_kernel(int32_t n_elems, int32_t n_nodes, int32_t n_parts, int32_t words, int32_t rank,
                            const int32_t *__restrict__ tets, const int32_t *__restrict__ epart, unsigned long long *mask,
                            uint32_t *key1, uint32_t *elem_key, int32_t *bad) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n_elems) return;
  const int32_t q = epart[e];
  if (q < 0 || q >= n_parts) {
    atomicMax(bad, 1);
    elem_key[e] = kNone;
    return;
  }
  elem_key[e] = q == rank ? (uint32_t)e : kNone;
  for (int c = 0; c < 4; ++c) {
    const int32_t v = tets[4 * e + c];
    if (v < 0 || v >= n_nodes) {
      atomicMax(bad, 2);
      continue;
    }
    atomicOr(&mask[(int64_t)v * words + (q >> 6)], 1ull << (q & 63));
    if (q == rank) atomicMin(&key1[v], (uint32_t)(4 * e + c));
  }
}

// per node: is it held by more than one part (-> Global_shared), and which is the lowest part other than mine
__global__ void holders_kernel(int32_t n_nodes, int32_t words, int32_t rank, const unsigned long long *__restrict__ mask,
                               uint32_t *gs_key, int32_t *lowest_other) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= n_nodes) return;
  int count = 0, low = -1;
  for (int w = 0; w < words; ++w) {
    unsigned long long m = mask[v * words + w];
    count += __popcll(m);
    if (w == (rank >> 6)) m &= ~(1ull << (rank & 63));
    if (low < 0 && m != 0) low = 64 * w + __ffsll((long long)m) - 1;
  }
  gs_key[v] = count > 1 ? (uint32_t)v : kNone;
  lowest_other[v] = low;
}

__global__ void second_key_kernel(int32_t n_elems, int32_t n_nodes, int32_t n_parts, int32_t rank, const int32_t *__restrict__ tets,
                                  const int32_t *__restrict__ epart, const uint32_t *__restrict__ key1,
                                  const int32_t *__restrict__ lowest_other, uint32_t *key2) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= n_elems) return;
  const int32_t q = epart[e];
  if (q == rank || q < 0 || q >= n_parts) return;
  for (int c = 0; c < 4; ++c) {
    const int32_t v = tets[4 * e + c];
    if (v < 0 || v >= n_nodes) continue;
    if (key1[v] != kNone && lowest_other[v] == q) atomicMin(&key2[v], (uint32_t)(4 * e + c));
  }
}

__global__ void shared_key_kernel(int32_t n_nodes, const uint32_t *__restrict__ key1, const int32_t *__restrict__ lowest_other,
                                  const uint32_t *__restrict__ key2, unsigned long long *key3) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= n_nodes) return;
  const bool cand = key1[v] != kNone && lowest_other[v] >= 0;
  key3[v] = cand ? ((unsigned long long)lowest_other[v] << 32) | key2[v] : kNone64;
}

// facets with all three nodes on x = 0 (|x| < tol, Data_prepare.py:131): first-seen key of their nodes
__global__ void clamp_key_kernel(int32_t n_facets, int32_t n_nodes, const int32_t *__restrict__ facets, const double *__restrict__ xyz,
                                 double tol, uint32_t *keyd, int32_t *bad) {
  const int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (f >= n_facets) return;
  int32_t v[3];
  bool on = true;
  for (int k = 0; k < 3; ++k) {
    v[k] = facets[3 * f + k];
    if (v[k] < 0 || v[k] >= n_nodes) {
      atomicMax(bad, 3);
      return;
    }
    on = on && fabs(xyz[3 * (int64_t)v[k]]) < tol;
  }
  if (on)
    for (int k = 0; k < 3; ++k) atomicMin(&keyd[v[k]], (uint32_t)(3 * f + k));
}

__global__ void iota_kernel(int32_t n, int32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)i;
}

template <typename K>
__global__ void count_valid_kernel(int32_t n, const K *__restrict__ keys, K none, int32_t *count) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const bool valid = i < n && keys[i] != none;
  const unsigned long long b = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}

// position[list[i]] = i
__global__ void scatter_positions_kernel(int32_t n, const int32_t *__restrict__ list, int32_t *position) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) position[list[i]] = (int32_t)i;
}

__global__ void local_cells_kernel(int32_t n_local_elems, const int32_t *__restrict__ elements, const int32_t *__restrict__ tets,
                                   const int32_t *__restrict__ local_of, int32_t *cells_local) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= 4 * (int64_t)n_local_elems) return;
  cells_local[i] = local_of[tets[4 * (int64_t)elements[i >> 2] + (i & 3)]];
}

__global__ void gather_kernel(int32_t n, const int32_t *__restrict__ list, const int32_t *__restrict__ table, int32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = table[list[i]];
}

// key of local node i: i if its global node is clamped
__global__ void local_clamp_key_kernel(int32_t n_local, const int32_t *__restrict__ nodes, const uint32_t *__restrict__ keyd,
                                       uint32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n_local) out[i] = keyd[nodes[i]] != kNone ? (uint32_t)i : kNone;
}

struct Scratch {
  std::vector<void *> p;
  template <typename T>
  hipError_t alloc(T **out, size_t count) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
    if (e == hipSuccess) p.push_back(q);
    *out = static_cast<T *>(q);
    return e;
  }
  ~Scratch() {
    for (void *q : p) (void)hipFree(q);
  }
};

#define TOPO_TRY(expr)              \
  do {                              \
    const hipError_t e_ = (expr);   \
    if (e_ != hipSuccess) return e_; \
  } while (0)

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)); }

// The indices 0..n-1 whose key is not `none`, ordered by key: *sorted (device, n entries, the valid ones first), *n_valid.
template <typename K>
hipError_t sorted_valid(Scratch &sc, const K *keys, int32_t n, K none, int32_t **sorted, int32_t *n_valid) {
  K *keys_s;
  int32_t *vals, *vals_s, *count;
  TOPO_TRY(sc.alloc(&keys_s, (size_t)n));
  TOPO_TRY(sc.alloc(&vals, (size_t)n));
  TOPO_TRY(sc.alloc(&vals_s, (size_t)n));
  TOPO_TRY(sc.alloc(&count, 1));
  *sorted = vals_s;
  *n_valid = 0;
  if (n == 0) return hipSuccess;
  TOPO_TRY(hipMemset(count, 0, sizeof(int32_t)));
  hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, nullptr, n, vals);
  hipLaunchKernelGGL((count_valid_kernel<K>), grid_for(n), dim3(256), 0, nullptr, n, keys, none, count);
  size_t tmp_bytes = 0;
  TOPO_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_s, vals, vals_s, n));
  char *tmp;
  TOPO_TRY(sc.alloc(&tmp, tmp_bytes));
  TOPO_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_s, vals, vals_s, n));
  TOPO_TRY(hipMemcpy(n_valid, count, sizeof(int32_t), hipMemcpyDeviceToHost));
  return hipSuccess;
}

hipError_t download(std::vector<int32_t> &dst, const int32_t *src, size_t n) {
  dst.resize(n);
  return n ? hipMemcpy(dst.data(), src, n * sizeof(int32_t), hipMemcpyDeviceToHost) : hipSuccess;
}

}  // namespace

hipError_t rank_topology(int device, int32_t n_nodes, int32_t n_elems, const int32_t *tets_host, const int32_t *epart_host,
                         int32_t rank, int32_t n_parts, const double *xyz_host, int32_t n_facets, const int32_t *facets_host,
                         double clamp_tol, RankTopology *out, std::string &err) {
  if (!out || n_nodes < 0 || n_elems < 0 || n_parts < 1 || rank < 0 || rank >= n_parts || (n_elems > 0 && (!tets_host || !epart_host)) ||
      n_facets < 0 || (n_facets > 0 && (!facets_host || !xyz_host))) {
    err = "saa_topology_build: bad argument";
    return hipErrorInvalidValue;
  }
  if (n_elems >= (1 << 30)) {
    err = "saa_topology_build: more than 2^30 elements";
    return hipErrorInvalidValue;
  }
  TOPO_TRY(hipSetDevice(device));
  Scratch sc;
  const int32_t words = (n_parts + 63) / 64;
  int32_t *tets, *epart, *lowest_other, *bad, *facets = nullptr;
  unsigned long long *mask, *key3;
  uint32_t *key1, *key2, *elem_key, *gs_key, *keyd;
  double *xyz = nullptr;
  TOPO_TRY(sc.alloc(&tets, 4 * (size_t)n_elems));
  TOPO_TRY(sc.alloc(&epart, (size_t)n_elems));
  TOPO_TRY(sc.alloc(&mask, (size_t)n_nodes * words));
  TOPO_TRY(sc.alloc(&key1, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&key2, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&key3, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&keyd, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&gs_key, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&lowest_other, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&elem_key, (size_t)n_elems));
  TOPO_TRY(sc.alloc(&bad, 1));
  if (n_elems) {
    TOPO_TRY(hipMemcpy(tets, tets_host, 4 * (size_t)n_elems * sizeof(int32_t), hipMemcpyHostToDevice));
    TOPO_TRY(hipMemcpy(epart, epart_host, (size_t)n_elems * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  TOPO_TRY(hipMemset(mask, 0, (size_t)n_nodes * words * sizeof(unsigned long long)));
  TOPO_TRY(hipMemset(key1, 0xff, (size_t)n_nodes * sizeof(uint32_t)));
  TOPO_TRY(hipMemset(key2, 0xff, (size_t)n_nodes * sizeof(uint32_t)));
  TOPO_TRY(hipMemset(keyd, 0xff, (size_t)n_nodes * sizeof(uint32_t)));
  TOPO_TRY(hipMemset(bad, 0, sizeof(int32_t)));
  if (n_elems)
    hipLaunchKernelGGL(mark_kernel, grid_for(n_elems), dim3(256), 0, nullptr, n_elems, n_nodes, n_parts, words, rank, tets, epart, mask,
                       key1, elem_key, bad);
  if (n_nodes)
    hipLaunchKernelGGL(holders_kernel, grid_for(n_nodes), dim3(256), 0, nullptr, n_nodes, words, rank, mask, gs_key, lowest_other);
  if (n_elems)
    hipLaunchKernelGGL(second_key_kernel, grid_for(n_elems), dim3(256), 0, nullptr, n_elems, n_nodes, n_parts, rank, tets, epart, key1,
                       lowest_other, key2);
  if (n_nodes)
    hipLaunchKernelGGL(shared_key_kernel, grid_for(n_nodes), dim3(256), 0, nullptr, n_nodes, key1, lowest_other, key2, key3);
  if (n_facets) {
    TOPO_TRY(sc.alloc(&facets, 3 * (size_t)n_facets));
    TOPO_TRY(sc.alloc(&xyz, 3 * (size_t)n_nodes));
    TOPO_TRY(hipMemcpy(facets, facets_host, 3 * (size_t)n_facets * sizeof(int32_t), hipMemcpyHostToDevice));
    TOPO_TRY(hipMemcpy(xyz, xyz_host, 3 * (size_t)n_nodes * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(clamp_key_kernel, grid_for(n_facets), dim3(256), 0, nullptr, n_facets, n_nodes, facets, xyz, clamp_tol, keyd, bad);
  }
  int32_t bad_host = 0;
  TOPO_TRY(hipMemcpy(&bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (bad_host) {
    err = bad_host == 1 ? "saa_topology_build: element part outside [0, n_parts)"
                        : (bad_host == 2 ? "saa_topology_build: element node outside [0, n_nodes)"
                                         : "saa_topology_build: facet node outside [0, n_nodes)");
    return hipErrorInvalidValue;
  }
  int32_t *elements, *nodes, *gshared, *shared, *dnodes, *dlocal;
  int32_t n_le = 0, n_ln = 0, n_gs = 0, n_sh = 0, n_d = 0, n_dl = 0;
  TOPO_TRY(sorted_valid<uint32_t>(sc, elem_key, n_elems, kNone, &elements, &n_le));
  TOPO_TRY(sorted_valid<uint32_t>(sc, key1, n_nodes, kNone, &nodes, &n_ln));
  TOPO_TRY(sorted_valid<uint32_t>(sc, gs_key, n_nodes, kNone, &gshared, &n_gs));
  TOPO_TRY(sorted_valid<unsigned long long>(sc, key3, n_nodes, kNone64, &shared, &n_sh));
  TOPO_TRY(sorted_valid<uint32_t>(sc, keyd, n_nodes, kNone, &dnodes, &n_d));
  // local numbering and what follows from it
  int32_t *local_of, *slot_of, *cells_local, *shared_local, *shared_slots;
  uint32_t *dl_key;
  TOPO_TRY(sc.alloc(&local_of, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&slot_of, (size_t)n_nodes));
  TOPO_TRY(sc.alloc(&cells_local, 4 * (size_t)n_le));
  TOPO_TRY(sc.alloc(&shared_local, (size_t)n_sh));
  TOPO_TRY(sc.alloc(&shared_slots, (size_t)n_sh));
  TOPO_TRY(sc.alloc(&dl_key, (size_t)n_ln));
  if (n_ln) hipLaunchKernelGGL(scatter_positions_kernel, grid_for(n_ln), dim3(256), 0, nullptr, n_ln, nodes, local_of);
  if (n_gs) hipLaunchKernelGGL(scatter_positions_kernel, grid_for(n_gs), dim3(256), 0, nullptr, n_gs, gshared, slot_of);
  if (n_le)
    hipLaunchKernelGGL(local_cells_kernel, grid_for(4 * (int64_t)n_le), dim3(256), 0, nullptr, n_le, elements, tets, local_of, cells_local);
  if (n_sh) {
    hipLaunchKernelGGL(gather_kernel, grid_for(n_sh), dim3(256), 0, nullptr, n_sh, shared, local_of, shared_local);
    hipLaunchKernelGGL(gather_kernel, grid_for(n_sh), dim3(256), 0, nullptr, n_sh, shared, slot_of, shared_slots);
  }
  if (n_ln) hipLaunchKernelGGL(local_clamp_key_kernel, grid_for(n_ln), dim3(256), 0, nullptr, n_ln, nodes, keyd, dl_key);
  TOPO_TRY(sorted_valid<uint32_t>(sc, dl_key, n_ln, kNone, &dlocal, &n_dl));
  TOPO_TRY(hipGetLastError());
  TOPO_TRY(download(out->elements, elements, n_le));
  TOPO_TRY(download(out->nodes, nodes, n_ln));
  TOPO_TRY(download(out->cells_local, cells_local, 4 * (size_t)n_le));
  TOPO_TRY(download(out->shared_nodes, shared, n_sh));
  TOPO_TRY(download(out->shared_local, shared_local, n_sh));
  TOPO_TRY(download(out->shared_slots, shared_slots, n_sh));
  TOPO_TRY(download(out->global_shared, gshared, n_gs));
  TOPO_TRY(download(out->dirichlet_nodes, dnodes, n_d));
  TOPO_TRY(download(out->dirichlet_local, dlocal, n_dl));
  return hipSuccess;
}

}  // namespace saa
