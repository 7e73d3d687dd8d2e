// Diagnostic: device-to-device copy rate of a few kernel shapes (which one does saa_device_copy_bandwidth use?).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ab/copy_bw tools/copy_bw.hip && tools/ab/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_one(const f4 *__restrict__ s, f4 *__restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_chunk(const f4 *__restrict__ s, f4 *__restrict__ d, int64_t n) {
  // every block copies one contiguous chunk of U*256 elements
  const int64_t base = (int64_t)blockIdx.x * (U * 256) + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int j = 0; j < U; ++j) v[j] = base + j * 256 < n ? (NT ? __builtin_nontemporal_load(s + base + j * 256) : s[base + j * 256]) : f4{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < U; ++j)
    if (base + j * 256 < n) {
      if (NT) __builtin_nontemporal_store(v[j], d + base + j * 256);
      else d[base + j * 256] = v[j];
    }
}
template <int U>
__global__ void __launch_bounds__(256) k_stride(const f4 *__restrict__ s, f4 *__restrict__ d, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = i + j * stride < n ? s[i + j * stride] : f4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i + j * stride < n) d[i + j * stride] = v[j];
  }
}

template <typename F>
double timeit(F launch, int64_t bytes, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  launch();
  hipEventRecord(a, nullptr);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b, nullptr);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
}

int main() {
  for (int64_t mb : {256, 1024, 4096}) {
    const int64_t bytes = mb << 20, n = bytes / 16;
    f4 *s, *d;
    if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipMemset(s, 0x3c, bytes);
    hipMemset(d, 0, bytes);
    const int reps = mb >= 4096 ? 5 : 10;
    printf("%5lld MiB: one/thread %.2f", (long long)mb, timeit([&] { hipLaunchKernelGGL(k_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  chunk4 %.2f", timeit([&] { hipLaunchKernelGGL((k_chunk<4, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  chunk8 %.2f", timeit([&] { hipLaunchKernelGGL((k_chunk<8, false>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  chunk4nt %.2f", timeit([&] { hipLaunchKernelGGL((k_chunk<4, true>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  chunk8nt %.2f", timeit([&] { hipLaunchKernelGGL((k_chunk<8, true>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  stride4x2048 %.2f", timeit([&] { hipLaunchKernelGGL((k_stride<4>), dim3(2048), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  stride4x8192 %.2f", timeit([&] { hipLaunchKernelGGL((k_stride<4>), dim3(8192), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  stride1x4096 %.2f", timeit([&] { hipLaunchKernelGGL((k_stride<1>), dim3(4096), dim3(256), 0, nullptr, s, d, n); }, bytes, reps));
    printf("  hipMemcpyDtoD %.2f TB/s\n", timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, nullptr); }, bytes, reps));
    hipFree(s);
    hipFree(d);
  }
  return 0;
}
