// Sustained v_mfma_f32_32x32x2_f32 rate on this chip (DVFS included): every wave issues N
// back-to-back MFMAs over NACC accumulators, W waves per SIMD.  Ceiling reference for the
// fp32 contraction kernels (cdna_hip_programming.md rule 10: measure a known-good reference).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd, float* d) {
  const int blocks = 256 * waves_per_simd;     // 4 waves per block -> one per SIMD per block
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    if (rep == 2) printf("NACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.2f GHz-equivalent at 64 FLOP/clk/SIMD)\n", NACC,
                         waves_per_simd, ms, flop / ms / 1e9, flop / ms / 1e9 / (1024 * 64.0) * 1e3 / 1e3);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<1>(1, d); run<4>(1, d); run<4>(2, d); run<4>(3, d); run<2>(4, d);
  return 0;
}
