// gfx950 set-up kernels: what the reference computes once, before the time loop, with dense (3N)^2 matrices on rank 0
// (/root/reference Data_prepare.py:147-154,175-176):
//   * lumped mass   = row sums of the consistent mass of Global_Assembly_no_bc (Tools/Mat_construction.py:199-231,
//                     Tools/commons.py:103-107); for linear tets the row sum is rho*V_e/4 per element and node
//                     (the 4-point rule integrates N_a N_b exactly, sum_b N_b = 1);
//   * F_pre         = pre-assembled un-ramped body force (0,-fz,-fz): (V_e/4)*(0,-fz,-fz) per element and node;
//   * Meshsize      = 2*min_edge/sqrt(24) over the rank's elements (Tools/commons.py:79-90), the CFL length.
// One pass over the elements (signed volume detJ/6, shortest squared edge) and one pass over the nodes.  The nodal sums
// are formed WITHOUT floating-point atomics: (node, element-corner) pairs are radix-sorted by node (stable), so every
// node adds its elements' V/4 in ascending element order - deterministic, and the order in which the host closed form
// (fem_setup.lumped_mass_and_load: np.bincount) accumulates too.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "saa_setup.h"

namespace saa {

__global__ void elem_volume_edge_kernel(int32_t n_elems, const double *__restrict__ xyz, const int32_t *__restrict__ tets,
                                        double *__restrict__ quarter_vol, int32_t *__restrict__ keys,
                                        int32_t *__restrict__ vals, unsigned long long *__restrict__ min_len2_bits) {
#pragma clang fp contract(off)
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double len2 = __builtin_huge_val();
  if (e < n_elems) {
    double p[4][3];
    for (int a = 0; a < 4; ++a) {
      const int32_t v = tets[4 * e + a];
      keys[4 * e + a] = v;
      vals[4 * e + a] = (int32_t)(4 * e + a);
      for (int c = 0; c < 3; ++c) p[a][c] = xyz[3 * (int64_t)v + c];
    }
    // signed volume detJ/6, J columns = edges from node 0 (Shape_function_Deriv.py:60-67, Mat_construction.py:93);
    // same expression order as fem_setup.signed_volumes (e1 . (e2 x e3))
    double e1[3], e2[3], e3[3];
    for (int c = 0; c < 3; ++c) {
      e1[c] = p[1][c] - p[0][c];
      e2[c] = p[2][c] - p[0][c];
      e3[c] = p[3][c] - p[0][c];
    }
    const double cx = e2[1] * e3[2] - e2[2] * e3[1];
    const double cy = e2[2] * e3[0] - e2[0] * e3[2];
    const double cz = e2[0] * e3[1] - e2[1] * e3[0];
    const double det = (e1[0] * cx + e1[1] * cy) + e1[2] * cz;
    quarter_vol[e] = (det / 6.0) / 4.0;
    // the six edges of commons.py:82-87
    const int pa[6] = {0, 1, 2, 1, 0, 0}, pb[6] = {1, 2, 3, 3, 3, 2};
    for (int k = 0; k < 6; ++k) {
      const double dx = p[pa[k]][0] - p[pb[k]][0], dy = p[pa[k]][1] - p[pb[k]][1], dz = p[pa[k]][2] - p[pb[k]][2];
      const double l2 = (dx * dx + dy * dy) + dz * dz;
      len2 = l2 < len2 ? l2 : len2;
    }
  }
  // non-negative doubles order like their bit patterns: wave minimum, then one integer atomic per wave
  unsigned long long bits = (unsigned long long)__double_as_longlong(len2);
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(bits, off, 64);
    bits = o < bits ? o : bits;
  }
  if ((threadIdx.x & 63) == 0) atomicMin(min_len2_bits, bits);
}

// offsets[v] = first position of node v in the sorted key array (offsets[n_nodes] = 4*n_elems)
__global__ void segment_offsets_kernel(int64_t n_pairs, int32_t n_nodes, const int32_t *__restrict__ sorted_keys,
                                       int64_t *__restrict__ offsets) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > n_pairs) return;
  const int32_t prev = i == 0 ? -1 : sorted_keys[i - 1];
  const int32_t cur = i == n_pairs ? n_nodes : sorted_keys[i];
  for (int32_t v = prev + 1; v <= cur; ++v) offsets[v] = i;
}

__global__ void nodal_fields_kernel(int32_t n_nodes, const int64_t *__restrict__ offsets, const int32_t *__restrict__ sorted_vals,
                                    const double *__restrict__ quarter_vol, double rho, double fz,
                                    double *__restrict__ lumped, double *__restrict__ fpre) {
#pragma clang fp contract(off)
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= n_nodes) return;
  double nodal = 0.0;
  for (int64_t i = offsets[v]; i < offsets[v + 1]; ++i) nodal += quarter_vol[sorted_vals[i] >> 2];  // ascending element order
  const double m = rho * nodal;
  lumped[3 * v] = m;
  lumped[3 * v + 1] = m;
  lumped[3 * v + 2] = m;
  fpre[3 * v] = nodal * 0.0;
  fpre[3 * v + 1] = nodal * (-fz);
  fpre[3 * v + 2] = nodal * (-fz);
}

namespace {
struct Scratch {
  void *p[16] = {};
  int n = 0;
  template <typename T>
  hipError_t alloc(T **out, size_t count) {
    void *q = nullptr;
    if (n >= 16) return hipErrorOutOfMemory;
    const hipError_t e = hipMalloc(&q, count ? count * sizeof(T) : sizeof(T));
    if (e == hipSuccess) p[n++] = q;
    *out = static_cast<T *>(q);
    return e;
  }
  ~Scratch() {
    for (int i = 0; i < n; ++i) (void)hipFree(p[i]);
  }
};
}  // namespace

#define SETUP_TRY(expr)            \
  do {                             \
    const hipError_t e_ = (expr);  \
    if (e_ != hipSuccess) return e_; \
  } while (0)

hipError_t setup_fields(int32_t n_nodes, int32_t n_elems, const double *xyz_host, const int32_t *tets_host, double rho, double fz,
                        double *lumped_host, double *fpre_host, double *min_edge_host) {
  Scratch sc;
  const int64_t n_pairs = 4 * static_cast<int64_t>(n_elems);
  double *xyz, *qvol, *lumped, *fpre;
  int32_t *tets, *keys, *vals, *keys_s, *vals_s;
  int64_t *offsets;
  unsigned long long *minbits;
  SETUP_TRY(sc.alloc(&xyz, 3 * static_cast<size_t>(n_nodes)));
  SETUP_TRY(sc.alloc(&tets, static_cast<size_t>(n_pairs)));
  SETUP_TRY(sc.alloc(&qvol, static_cast<size_t>(n_elems)));
  SETUP_TRY(sc.alloc(&keys, static_cast<size_t>(n_pairs)));
  SETUP_TRY(sc.alloc(&vals, static_cast<size_t>(n_pairs)));
  SETUP_TRY(sc.alloc(&keys_s, static_cast<size_t>(n_pairs)));
  SETUP_TRY(sc.alloc(&vals_s, static_cast<size_t>(n_pairs)));
  SETUP_TRY(sc.alloc(&offsets, static_cast<size_t>(n_nodes) + 1));
  SETUP_TRY(sc.alloc(&minbits, 1));
  SETUP_TRY(hipMemcpy(xyz, xyz_host, 3 * static_cast<size_t>(n_nodes) * sizeof(double), hipMemcpyHostToDevice));
  if (n_elems > 0) SETUP_TRY(hipMemcpy(tets, tets_host, static_cast<size_t>(n_pairs) * sizeof(int32_t), hipMemcpyHostToDevice));
  SETUP_TRY(hipMemset(minbits, 0x7f, sizeof(unsigned long long)));  // 0x7f7f... : a huge finite double
  const int threads = 256;
  if (n_elems > 0)
    hipLaunchKernelGGL(elem_volume_edge_kernel, dim3((unsigned)((n_elems + threads - 1) / threads)), dim3(threads), 0, nullptr,
                       n_elems, xyz, tets, qvol, keys, vals, minbits);
  // stable sort of the (node, 4*element + corner) pairs by node
  if (n_pairs > 0) {
    size_t tmp_bytes = 0;
    int end_bit = 1;
    while (end_bit < 31 && (1ll << end_bit) < n_nodes) ++end_bit;
    SETUP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_s, vals, vals_s, static_cast<int>(n_pairs), 0, end_bit));
    char *tmp = nullptr;  // (owned by `sc`: freed on every exit path)
    SETUP_TRY(sc.alloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    SETUP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_s, vals, vals_s, static_cast<int>(n_pairs), 0, end_bit));
    SETUP_TRY(hipDeviceSynchronize());
  }
  hipLaunchKernelGGL(segment_offsets_kernel, dim3((unsigned)((n_pairs + 1 + threads - 1) / threads)), dim3(threads), 0, nullptr,
                     n_pairs, n_nodes, keys_s, offsets);
  SETUP_TRY(sc.alloc(&lumped, 3 * static_cast<size_t>(n_nodes)));
  SETUP_TRY(sc.alloc(&fpre, 3 * static_cast<size_t>(n_nodes)));
  hipLaunchKernelGGL(nodal_fields_kernel, dim3((unsigned)((n_nodes + threads - 1) / threads)), dim3(threads), 0, nullptr, n_nodes,
                     offsets, vals_s, qvol, rho, fz, lumped, fpre);
  SETUP_TRY(hipGetLastError());
  SETUP_TRY(hipDeviceSynchronize());
  if (lumped_host) SETUP_TRY(hipMemcpy(lumped_host, lumped, 3 * static_cast<size_t>(n_nodes) * sizeof(double), hipMemcpyDeviceToHost));
  if (fpre_host) SETUP_TRY(hipMemcpy(fpre_host, fpre, 3 * static_cast<size_t>(n_nodes) * sizeof(double), hipMemcpyDeviceToHost));
  if (min_edge_host) {
    unsigned long long bits = 0;
    SETUP_TRY(hipMemcpy(&bits, minbits, sizeof(bits), hipMemcpyDeviceToHost));
    double len2;
    static_assert(sizeof(len2) == sizeof(bits), "double is 64 bits");
    __builtin_memcpy(&len2, &bits, sizeof(len2));
    *min_edge_host = n_elems > 0 ? __builtin_sqrt(len2) : 0.0;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// Device-to-device copy rate of this GPU: the practical HBM ceiling next to the nominal 8 TB/s.
// 16 bytes per lane (global_load_dwordx4 / global_store_dwordx4), ONE element per thread, as many workgroups as it takes: the
// shape that reaches the 6.2 TB/s MI355X_MICROARCH.md quotes for a float4 copy (measured on MI355X, 1 GiB -> 1 GiB, TB/s:
// this 6.22; four elements per thread from one contiguous chunk 5.99, eight 5.60; grid-stride loops of 2048-8192
// workgroups 4.4-4.9; hipMemcpyDtoD 5.49; torch's copy_ 4.8-5.1 - tools/copy_bw.hip, profiles/r03_copy_shapes.txt).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy16_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

hipError_t copy_bandwidth(int device, int64_t n_bytes, int reps, double *bytes_per_s) {
  Scratch sc;
  const int64_t n = n_bytes / 16;
  double2 *a, *b;
  (void)device;
  SETUP_TRY(sc.alloc(&a, static_cast<size_t>(n)));
  SETUP_TRY(sc.alloc(&b, static_cast<size_t>(n)));
  SETUP_TRY(hipMemset(a, 0x3c, static_cast<size_t>(n) * 16));  // finite, non-zero doubles
  hipEvent_t e0, e1;
  SETUP_TRY(hipEventCreate(&e0));
  SETUP_TRY(hipEventCreate(&e1));
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(copy16_kernel, grid, dim3(256), 0, nullptr, a, b, n);  // warm
  hipError_t e = hipEventRecord(e0, nullptr);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy16_kernel, grid, dim3(256), 0, nullptr, a, b, n);
  if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e == hipSuccess) e = hipGetLastError();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e == hipSuccess) *bytes_per_s = ms > 0.f ? 2.0 * 16.0 * static_cast<double>(n) * reps / (ms * 1e-3) : 0.0;
  return e;
}

}  // namespace saa
