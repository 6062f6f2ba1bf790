// Row samplers shared by the sampling kernels (wn_elem.hip) and the generation head kernel (wn_gen.hip): one wave per row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

// ------------------------------------------------------------------------------------------
// wave-per-row helpers
// Reductions over the 64 lanes on the DPP cross-lane operands of the VALU (no LDS round trips: __shfl_xor is a ds_bpermute
// with its own address and wait, six in a row per reduction): quad butterflies, then the two mirror permutations leave
// every lane with the sum / max of its row of 16; the four row results are read as scalars.  Every lane returns the result.
// (All 64 lanes must be active, as with the shuffles.)
#define WN_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
__device__ __forceinline__ float wn_wave_max(float v) {
  v = fmaxf(v, WN_DPP_F(v, 0xB1));                  // quad_perm [1, 0, 3, 2]
  v = fmaxf(v, WN_DPP_F(v, 0x4E));                  // quad_perm [2, 3, 0, 1]
  v = fmaxf(v, WN_DPP_F(v, 0x141));                 // row_half_mirror
  v = fmaxf(v, WN_DPP_F(v, 0x140));                 // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wn_wave_sum(float v) {
  v += WN_DPP_F(v, 0xB1);
  v += WN_DPP_F(v, 0x4E);
  v += WN_DPP_F(v, 0x141);
  v += WN_DPP_F(v, 0x140);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
// inclusive prefix sum over the lanes: shifts by 1, 2, 4, 8 inside the rows of 16 (lanes without a source add zero), then
// the last lane of rows 0 / 2 into rows 1 / 3 and lane 31 into the upper half
#define WN_DPP_Z(v, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wn_wave_scan_incl(float v) {
  v += WN_DPP_Z(v, 0x111, 0xf);                     // row_shr:1
  v += WN_DPP_Z(v, 0x112, 0xf);                     // row_shr:2
  v += WN_DPP_Z(v, 0x114, 0xf);                     // row_shr:4
  v += WN_DPP_Z(v, 0x118, 0xf);                     // row_shr:8
  v += WN_DPP_Z(v, 0x142, 0xa);                     // row_bcast:15 into rows 1, 3
  v += WN_DPP_Z(v, 0x143, 0xc);                     // row_bcast:31 into rows 2, 3
  return v;
}

// Philox4x32-10 (Salmon et al. 2011), counter = (row, offset), key = seed
__device__ __forceinline__ void wn_philox(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float wn_u01(uint32_t r) {   // (0,1), 24-bit
  return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// inverse-CDF categorical draw from the (unnormalised) probabilities p[0..C) of one row per wave
// (p may live in global memory or in LDS); every lane returns the drawn class
template <typename P>
__device__ __forceinline__ int wn_draw_cat_row(P p, int C, int lane, int64_t row, uint64_t seed, uint64_t offset) {
  const int per = (C + 63) / 64;               // contiguous chunk per lane
  const int j0 = lane * per;
  float loc = 0.f;
  for (int j = j0; j < min(C, j0 + per); ++j) loc += fmaxf(p[j], 0.f);
  const float incl = wn_wave_scan_incl(loc);   // inclusive scan over lanes
  const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
  uint32_t r[4];
  wn_philox((uint64_t)row, offset, seed, r);
  const float target = wn_u01(r[0]) * total;
  const unsigned long long hit = __ballot(incl > target);
  const int sel_lane = hit ? __builtin_ctzll(hit) : 63;
  int result = C - 1;
  if (lane == sel_lane) {
    float run = incl - loc;
    result = min(C, j0 + per) - 1;
    for (int j = j0; j < min(C, j0 + per); ++j) {
      run += fmaxf(p[j], 0.f);
      if (run > target) { result = j; break; }
    }
  }
  return __shfl(result, sel_lane);
}


// Categorical head, deterministic (src/model.py:393-421 with deterministic sampling): softmax -> arg max -> sample value.
// The arithmetic is that of wn_softmax_kernel followed by wn_sample_det_cat_kernel (same lane assignment, same
// reductions), so the result is the same sample.  Every lane returns it.
__device__ __forceinline__ float wn_cat_det_row(const float* l, int C, int lane, float inv_lv) {
  float m = -INFINITY;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, l[j]);
  m = wn_wave_max(m);
  float z = 0.f;
  for (int j = lane; j < C; j += 64) z += expf(l[j] - m);
  z = wn_wave_sum(z);
  const float inv = 1.0f / z;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < C; j += 64) {
    const float v = expf(l[j] - m) * inv;        // the probability wn_softmax_kernel would have stored
    if (v > best) { best = v; bi = j; }          // strictly greater keeps the first maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  return (float)bi * inv_lv - 1.0f;
}

// Categorical head, stochastic draw straight from the logits: the probabilities are those of wn_softmax_kernel (same lane
// assignment, same reductions), kept in the LDS row q[0..C) instead of a (rows, C) tensor in HBM, so the drawn class is
// the one sample_waveform(softmax(logits)) draws.  Every lane returns the sample value.
__device__ __forceinline__ float wn_cat_rand_row(const float* l, int C, int lane, float* q, int64_t row, uint64_t seed,
                                                 uint64_t offset, float inv_lv) {
  if (C <= 256) {
    // one read of the row, one exp per class (element k of a lane = class lane + 64 k: the same per-lane
    // order of the max / sum as the loops below)
    float v[4], e[4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = lane + 64 * k < C ? l[lane + 64 * k] : -INFINITY;
      m = fmaxf(m, v[k]);
    }
    m = wn_wave_max(m);
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < C) { e[k] = expf(v[k] - m); z += e[k]; }
    z = wn_wave_sum(z);
    const float inv = 1.0f / z;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < C) q[lane + 64 * k] = e[k] * inv;
  } else {
    float m = -INFINITY;
    for (int j = lane; j < C; j += 64) m = fmaxf(m, l[j]);
    m = wn_wave_max(m);
    float z = 0.f;
    for (int j = lane; j < C; j += 64) z += expf(l[j] - m);
    z = wn_wave_sum(z);
    const float inv = 1.0f / z;
    for (int j = lane; j < C; j += 64) q[j] = expf(l[j] - m) * inv;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int result = wn_draw_cat_row((const float*)q, C, lane, row, seed, offset);
  return (float)result * inv_lv - 1.0f;
}

// Mixture heads (logistic / gaussian), one row per thread: src/model.py:423-503.  p = [M logit weights | M means | M log scales].
__device__ __forceinline__ float wn_mix_det_row(const float* p, int M) {
  int bi = 0;
  float best = p[0];
  for (int k = 1; k < M; ++k) if (p[k] > best) { best = p[k]; bi = k; }
  return fminf(fmaxf(p[M + bi], -1.0f), 1.0f);
}
__device__ __forceinline__ float wn_mix_rand_row(const float* p, int M, int kind, int64_t row, uint64_t seed, uint64_t offset) {
  uint32_t r[4];
  wn_philox((uint64_t)row, offset, seed, r);
  float wm = -INFINITY;
  for (int k = 0; k < M; ++k) wm = fmaxf(wm, p[k]);
  float wz = 0.f;
  for (int k = 0; k < M; ++k) wz += expf(p[k] - wm);
  const float target = wn_u01(r[0]) * wz;
  int sel = M - 1;
  float run = 0.f;
  for (int k = 0; k < M; ++k) { run += expf(p[k] - wm); if (run > target) { sel = k; break; } }
  const float mu = p[M + sel], sc = expf(p[2 * M + sel]);
  float v;
  if (kind == 1) {                       // logistic: mu + s (ln z - ln(1-z))     src/model.py:463-483
    const float zz = wn_u01(r[1]);
    v = mu + sc * (logf(zz) - logf(1.0f - zz));
  } else {                               // gaussian: mu + s n                   src/model.py:423-443
    const float u1 = wn_u01(r[1]), u2 = wn_u01(r[2]);
    v = mu + sc * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  }
  return fminf(fmaxf(v, -1.0f), 1.0f);
}
