// Probe: is a chain of v_mfma_f32_16x16x16_f16 over the same operands bit-identical to the chain of
// v_mfma_f32_32x32x16_f16 the forward kernels run (same K = 16 per instruction, fp32 accumulate)?  And what does a
// DEPENDENT chain of each cost per instruction?  (Generation kernels are latency chains of dependent products; a
// smaller tile with the same arithmetic would shorten them.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// D = A (32 x K) * B (K x 32), operands fp16 (already rounded), one 32x32 tile, chain over K / 16 k-steps
__global__ void big(const _Float16* A, const _Float16* B, float* D, int K, long long* cyc) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int k0 = 0; k0 < K; k0 += 16) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[i * K + k0 + 8 * h + j]; b[j] = B[(k0 + 8 * h + j) * 32 + i]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
  if (lane == 0) cyc[0] = t1 - t0;
}
// the same product as four 16 x 16 tiles: tile (ti, tj) by this wave one after the other; per tile a chain over K / 16
__global__ void small(const _Float16* A, const _Float16* B, float* D, int K, long long* cyc) {
  const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
  long long tt = 0;
  for (int ti = 0; ti < 2; ++ti)
    for (int tj = 0; tj < 2; ++tj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const long long t0 = __builtin_amdgcn_s_memtime();
      for (int k0 = 0; k0 < K; k0 += 16) {
        h4 a, b;
        for (int j = 0; j < 4; ++j) { a[j] = A[(16 * ti + n) * K + k0 + 4 * g + j]; b[j] = B[(k0 + 4 * g + j) * 32 + 16 * tj + n]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
      }
      tt += __builtin_amdgcn_s_memtime() - t0;
      for (int r = 0; r < 4; ++r) D[(16 * ti + 4 * g + r) * 32 + 16 * tj + n] = acc[r];
    }
  if (lane == 0) cyc[1] = tt / 4;
}
// dependent chains on register operands only (latency per instruction): n instructions, timed with events from the host
__global__ void lat32(int n, float* sink) {
  h8 a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(0.001f * (threadIdx.x + j)); b8[j] = (_Float16)(0.002f * j); }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c, 0, 0, 0);
  }
  sink[threadIdx.x] = c[0];
}
__global__ void lat16(int n, float* sink) {
  h4 a4, b4;
  for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)(0.001f * (threadIdx.x + j)); b4[j] = (_Float16)(0.002f * j); }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);
  }
  sink[threadIdx.x] = c[0];
}

int main() {
  const int K = 256;
  _Float16 *A, *B; float *D1, *D2, *sink; long long* cyc;
  hipMallocManaged(&A, 32 * K * 2); hipMallocManaged(&B, K * 32 * 2);
  hipMallocManaged(&D1, 4096); hipMallocManaged(&D2, 4096); hipMallocManaged(&sink, 256); hipMallocManaged(&cyc, 64);
  int bad_total = 0;
  for (int trial = 0; trial < 20; ++trial) {
    srand(trial + 1);
    const double sa = trial % 3 == 0 ? 1.0 : (trial % 3 == 1 ? 1e-3 : 30.0);
    for (int n = 0; n < 32 * K; ++n) { A[n] = (_Float16)((rand() / (double)RAND_MAX * 2 - 1) * sa); B[n] = (_Float16)((rand() / (double)RAND_MAX * 2 - 1) * 0.5); }
    hipLaunchKernelGGL(big, dim3(1), dim3(64), 0, 0, A, B, D1, K, cyc);
    hipLaunchKernelGGL(small, dim3(1), dim3(64), 0, 0, A, B, D2, K, cyc);
    hipDeviceSynchronize();
    int bad = 0; double mx = 0;
    for (int n = 0; n < 1024; ++n) if (memcmp(&D1[n], &D2[n], 4) != 0) { ++bad; mx = fmax(mx, fabs((double)D1[n] - D2[n])); }
    bad_total += bad;
    if (trial < 3 || bad) printf("trial %d scale %g: %d of 1024 elements differ (max |diff| %.3g, |D| ~ %.3g)\n", trial, sa, bad, mx, fabs((double)D1[5]));
  }
  printf("total differing elements over 20 trials: %d\n", bad_total);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const int n = 4000000;
  hipLaunchKernelGGL(lat32, dim3(1), dim3(64), 0, 0, 800, sink);
  hipLaunchKernelGGL(lat16, dim3(1), dim3(64), 0, 0, 800, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(lat32, dim3(1), dim3(64), 0, 0, n, sink);
  hipEventRecord(e1, 0);
  hipLaunchKernelGGL(lat16, dim3(1), dim3(64), 0, 0, n, sink);
  hipEventRecord(e2, 0);
  hipDeviceSynchronize();
  float m32 = 0, m16 = 0; hipEventElapsedTime(&m32, e0, e1); hipEventElapsedTime(&m16, e1, e2);
  printf("dependent chain, register operands, one wave: 32x32x16 %.2f ns per instruction, 16x16x16 %.2f ns per instruction\n",
         m32 * 1e6 / n, m16 * 1e6 / n);
  return 0;
}
