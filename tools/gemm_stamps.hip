// Diagnostic: where the cycles of the predictor's input-projection GEMM go.  Compiles csrc/saa_predictor.hip with in-kernel
// stamps (-DSAA_GEMM_STAMPS) and runs GEMM 1 of the config-4 shape (3000 x 9126 fp64 history, 400 gate rows) on random data.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DSAA_GEMM_STAMPS tools/gemm_stamps.hip -o tools/ab/gemm_stamps
#include "../synchronization_avoiding_algorithms_amd/csrc/saa_predictor.hip"

#include <cstdio>
#include <random>

using namespace saa;

int main() {
  const int M = 3000, N = 400, K = 9126, ldb = 9152;
  std::mt19937 rng(1);
  std::uniform_real_distribution<double> u(-1e-3, 1e-3);
  std::vector<double> A((size_t)M * K);
  std::vector<float> B((size_t)N * ldb, 0.f);
  for (auto &v : A) v = u(rng);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) B[(size_t)n * ldb + k] = (float)u(rng) * 100.f;
  double *dA;
  float *dB, *dC;
  unsigned long long *dS;
  int S, kps;
  pick_splits(M, N, K, 256, &S, &kps);
  const dim3 grid((M + kBM - 1) / kBM, (N + kBN - 1) / kBN, S);
  const size_t n_wg = (size_t)grid.x * grid.y * grid.z;
  hipMalloc(&dA, A.size() * 8);
  hipMalloc(&dB, B.size() * 4);
  hipMalloc(&dC, (size_t)S * M * N * 4);
  hipMalloc(&dS, n_wg * 32 * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dS, 0, n_wg * 32 * 8);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)kGemmLds);
  GemmArgs g{};
  g.A = dA; g.lda = K; g.B = dB; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.k_per_split = kps;
  g.smax = 1e-3; g.sden = 2e-3; g.srcp = 1.0 / g.sden; g.Cpart = dC; g.ldc = N; g.stamps = dS;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((gemm_nt_kernel<true, false>), grid, dim3(kGemmThreads), kGemmLds, 0, g);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("launch %d: %.1f us, %.1f TFLOP/s (grid %u x %u x %u, %d k per split)\n", rep, ms * 1e3,
           2.0 * M * N * K / (ms * 1e-3) / 1e12, grid.x, grid.y, grid.z, kps);
  }
  std::vector<unsigned long long> T(n_wg * 32);
  hipMemcpy(T.data(), dS, T.size() * 8, hipMemcpyDeviceToHost);
  const char *names[6] = {"scale, store to LDS", "barrier 1", "request the chunk after",
                          "fragment reads + 104 MFMAs", "barrier 2", "wait for the next chunk"};
  double sum[6] = {0, 0, 0, 0, 0, 0};
  for (size_t w = 0; w < n_wg * 4; ++w)
    for (int j = 0; j < 6; ++j) sum[j] += (double)T[(w / 4) * 32 + (w % 4) * 8 + j];
  const double chunks = (double)((kps + kKC - 1) / kKC), waves = (double)n_wg * 4;
  double tot = 0;
  for (int j = 0; j < 6; ++j) tot += sum[j];
  for (int j = 0; j < 6; ++j)
    printf("%-46s %8.0f cycles per chunk and wave (%4.1f %%)\n", names[j], sum[j] / waves / chunks, 100 * sum[j] / tot);
  printf("%-46s %8.0f\n", "sum", tot / waves / chunks);
  return 0;
}
