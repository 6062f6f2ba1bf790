// ds_read_b64_tr_b16 lane map check: hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/tr_probe && tools/tr_probe
// LDS tile [time][64 channels] fp16, value = 100 * time + channel.  Lane l supplies the address of row (8h + q), columns
// 16 g + 4 p .. + 3 (h = l >> 5, g = (l >> 4) & 1, q = (l & 15) >> 2, p = l & 3) and must receive times 8h .. 8h + 3 of channel
// (l & 31): the A / B operand of v_mfma_f32_32x32x16_f16 with time as the contraction index.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (_Float16)(float)(100 * (i / 64) + (i % 64));
  __syncthreads();
  const int l = threadIdx.x;
  const int q = (l & 15) >> 2, p = l & 3;
  const _Float16* addr = lds + (8 * (l >> 5) + q) * 64 + 16 * ((l >> 4) & 1) + 4 * p;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
  h4 hv = *(h4*)&v;
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)hv[e];
}
int main() {
  float* d; float h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      const float want = 100.f * (8 * (l >> 5) + e) + (l & 31);
      if (h[l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %.0f want %.0f\n", l, e, h[l * 4 + e], want); ++bad; }
    }
  printf("tr_probe: %d mismatches\n", bad);
  return bad != 0;
}
