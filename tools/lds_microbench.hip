// Diagnostic microbenchmark (not product code): LDS-array cost per wave-instruction on gfx950 of the
// operations the fused step kernel leans on: ds_add_f64 under different address patterns, ds_read_b64 /
// ds_read2_b64 / ds_read_b128 gathers, ds_write_b64.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_microbench.hip -o /tmp/lds_mb && /tmp/lds_mb
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) double lds_double;

constexpr int kIters = 2000;
constexpr int kUnroll = 12;

// mode: 0 ds_add_f64, 1 ds_read_b64, 2 ds_read_b128 (double2), 3 ds_write_b64, 4 RMW (read+add+write)
template <int MODE>
__global__ void bench(const int *__restrict__ idx, double *out, int n_slots, long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < n_slots; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  int my[kUnroll];
  for (int j = 0; j < kUnroll; ++j) my[j] = idx[(blockIdx.x % 8) * blockDim.x * kUnroll + j * blockDim.x + tid];
  double acc = 0.0, v = 1.0 + tid;
  long long t0 = clock64();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      if (MODE == 0) {
        __builtin_amdgcn_ds_atomic_fadd_f64((lds_double *)(lds + my[j]), v);
      } else if (MODE == 1) {
        acc += lds[my[j]];
      } else if (MODE == 2) {
        const double2 t = *reinterpret_cast<const double2 *>(lds + (my[j] & ~1));
        acc += t.x + t.y;
      } else if (MODE == 3) {
        lds[my[j]] = v + j;
      } else {
        lds[my[j]] += v;
      }
    }
    if (MODE == 1 || MODE == 2) {
      v += acc * 1e-30;
#pragma unroll
      for (int j = 0; j < kUnroll; ++j) my[j] = (my[j] + 64) & (n_slots - 1);  // same bank, new address: no hoisting
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = acc + lds[tid % n_slots];
}

struct Pattern {
  const char *name;
  std::vector<int> idx;
};

int main() {
  const int threads = 512, blocks = 512, n_slots = 2048;  // 16 KiB of doubles per block
  std::vector<Pattern> pats;
  auto make = [&](const char *name, auto f) {
    Pattern p{name, std::vector<int>(8 * threads * kUnroll)};
    for (int b = 0; b < 8; ++b)
      for (int j = 0; j < kUnroll; ++j)
        for (int t = 0; t < threads; ++t) p.idx[(b * kUnroll + j) * threads + t] = f(b, j, t) % n_slots;
    pats.push_back(p);
  };
  srand(1);
  make("lane-contiguous (8 B stride)      ", [](int, int j, int t) { return t + 64 * j; });
  make("24 B stride (3*lane + j%3)        ", [](int, int j, int t) { return 3 * t + j % 3 + 7 * j; });
  make("48 B stride (6*lane)              ", [](int, int j, int t) { return 6 * t + j; });
  make("random distinct-ish               ", [](int, int, int) { return rand(); });
  make("random node*3+c (24 B records)    ", [](int, int j, int) { return 3 * (rand() % 372) + j % 3; });
  make("6 lanes share one address         ", [](int, int j, int t) { return 3 * (t / 6 + 11 * j) + j % 3; });
  make("2 lanes share one address         ", [](int, int j, int t) { return 3 * (t / 2 + 11 * j) + j % 3; });
  make("all 64 lanes one address          ", [](int, int j, int t) { return (t / 64) * 8 + j; });

  int *d_idx;
  double *d_out;
  long long *d_cyc;
  hipMalloc(&d_idx, 8 * threads * kUnroll * sizeof(int));
  hipMalloc(&d_out, blocks * threads * sizeof(double));
  hipMalloc(&d_cyc, blocks * sizeof(long long));
  std::vector<long long> cyc(blocks);
  const char *modes[] = {"ds_add_f64 ", "ds_read_b64", "ds_read_b128", "ds_write_b64", "rmw b64    "};
  for (auto &p : pats) {
    hipMemcpy(d_idx, p.idx.data(), p.idx.size() * sizeof(int), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 5; ++mode) {
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      auto launch = [&]() {
        switch (mode) {
          case 0: hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(threads), n_slots * 8, 0, d_idx, d_out, n_slots, d_cyc); break;
          case 1: hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(threads), n_slots * 8, 0, d_idx, d_out, n_slots, d_cyc); break;
          case 2: hipLaunchKernelGGL(bench<2>, dim3(blocks), dim3(threads), n_slots * 8, 0, d_idx, d_out, n_slots, d_cyc); break;
          case 3: hipLaunchKernelGGL(bench<3>, dim3(blocks), dim3(threads), n_slots * 8, 0, d_idx, d_out, n_slots, d_cyc); break;
          default: hipLaunchKernelGGL(bench<4>, dim3(blocks), dim3(threads), n_slots * 8, 0, d_idx, d_out, n_slots, d_cyc); break;
        }
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(a);
      launch();
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
      // 512 blocks x 8 waves on 256 CUs = 16 waves per CU; wave-instructions per CU:
      const double winstr_per_cu = 16.0 * kIters * kUnroll;
      const double ns_per = ms * 1e6 / winstr_per_cu;
      printf("%s | %s | %8.3f ms | %6.2f ns per wave-instr per CU (~%5.1f cyc @2.1GHz)\n", p.name, modes[mode], ms,
             ns_per, ns_per * 2.1);
    }
  }
  return 0;
}
