// Probe: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs, and how accurate is the
// hi/lo split (a = a_hi + a_lo in fp16, 3 products, fp32 accumulate) against an fp64 dot product?
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// D[i][j] = sum_k A[i][k] B[k][j], K = 16: lane l holds A[i=l&31][k=8h+jj], B[k=8h+jj][j=l&31]
__global__ void probe(const float* A, const float* B, float* D3, float* D1, int K) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  f32x16 acc3, acc1;
  for (int r = 0; r < 16; ++r) { acc3[r] = 0.f; acc1[r] = 0.f; }
  for (int k0 = 0; k0 < K; k0 += 16) {
    h8 ah, al, bh, bl;
    for (int jj = 0; jj < 8; ++jj) {
      const float a = A[i * K + k0 + 8 * h + jj], b = B[(k0 + 8 * h + jj) * 32 + i];
      const _Float16 a1 = (_Float16)a, b1 = (_Float16)b;
      ah[jj] = a1; al[jj] = (_Float16)(a - (float)a1);
      bh[jj] = b1; bl[jj] = (_Float16)(b - (float)b1);
    }
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc3, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    D3[row * 32 + i] = acc3[r];
    D1[row * 32 + i] = acc1[r];
  }
}

int main() {
  const int K = 128;
  float *A, *B, *D3, *D1;
  hipMallocManaged(&A, 32 * K * 4); hipMallocManaged(&B, K * 32 * 4);
  hipMallocManaged(&D3, 1024 * 4); hipMallocManaged(&D1, 1024 * 4);
  const double scales[] = {1.0, 1e-2, 1e-4, 3e-6};
  for (double sc : scales) {
    srand(1);
    for (int n = 0; n < 32 * K; ++n) { A[n] = (float)((rand() / (double)RAND_MAX * 2 - 1) * sc); B[n] = (float)((rand() / (double)RAND_MAX * 2 - 1) * 0.2); }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, D3, D1, K);
    hipDeviceSynchronize();
    double e3 = 0, e1 = 0, ef = 0, mx = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0; float f = 0.f;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * (double)B[k * 32 + j]; f = fmaf(A[i * K + k], B[k * 32 + j], f); }
      e3 = fmax(e3, fabs(D3[i * 32 + j] - ref)); e1 = fmax(e1, fabs(D1[i * 32 + j] - ref)); ef = fmax(ef, fabs((double)f - ref));
      mx = fmax(mx, fabs(ref));
    }
    printf("scale %.0e: max|ref| %.3e  err split3 %.3e (rel %.2e)  err fp16x1 %.3e  err fp32-fma %.3e\n", sc, mx, e3, e3 / mx, e1, ef);
  }
  // subnormal probe: a = 2^-20 (fp16 subnormal), b = 1: product must be 2^-20, not 0
  for (int n = 0; n < 32 * K; ++n) { A[n] = 0.f; B[n] = 0.f; }
  A[0] = ldexpf(1.f, -20); B[0] = 1.0f;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, D3, D1, 16);
  hipDeviceSynchronize();
  printf("subnormal input 2^-20 * 1 -> %.9e (expected %.9e): %s\n", D1[0], ldexp(1.0, -20), D1[0] != 0.f ? "KEPT" : "FLUSHED");
  return 0;
}
