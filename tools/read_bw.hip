// Pure-read bandwidth probe: how fast can the CUs pull a 1 GiB buffer, as a function of the bytes each wave keeps in
// flight and of the waves per CU?   hipcc --offload-arch=gfx950 -O3 tools/read_bw.hip -o tools/read_bw && tools/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(512) void read_kernel(const f32x4* __restrict__ src, size_t n16, float* out) {
  // grid-stride over 1 KiB wave blocks; U blocks in flight per wave
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  f32x4 acc = {0, 0, 0, 0};
  const size_t nblk = n16 / 64;
  for (size_t b = wave * U; b + U <= nblk; b += nwaves * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[(b + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int U>
static void run(const f32x4* d, size_t n16, float* out, int wgs, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((read_kernel<U>), dim3(wgs), dim3(threads), 0, 0, d, n16, out);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((read_kernel<U>), dim3(wgs), dim3(threads), 0, 0, d, n16, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("in flight %2d KiB/wave, %4d WGs x %3d threads: %7.1f GB/s\n", U, wgs, threads, 5.0 * n16 * 16 / (ms * 1e-3) / 1e9);
}
int main() {
  const size_t bytes = 1ull << 30, n16 = bytes / 16;
  f32x4* d; float* out;
  hipMalloc(&d, bytes); hipMalloc(&out, 4);
  hipMemset(d, 0, bytes);
  for (int threads : {256, 512}) for (int wgs : {256, 512, 1024, 2048}) {
    run<1>(d, n16, out, wgs, threads); run<2>(d, n16, out, wgs, threads); run<4>(d, n16, out, wgs, threads); run<8>(d, n16, out, wgs, threads);
  }
  return 0;
}
