// Weight gradient of one convolution tap as a staged workgroup kernel:
//   dW[k][n] = sum_t X[t - shift][k] * G[t][n],   db[n] = sum_t G[t][n]     (X[t - shift] = 0 for t < shift)
// One workgroup per (job, utterance, time range); operands staged ONCE per 16-step chunk: every thread
// fetches 8 consecutive time steps of one X channel and of one G channel (lanes = consecutive channels,
// compile-time row strides -> one base address + immediate offsets), splits them into fp16 hi | lo and
// stores 16 bytes per plane into a channel-major LDS stage [channel][16 t]; the (32 KT) x (32 NT) output
// block is tiled over the waves (2 x 2 tiles each) and every MFMA fragment (time = the MFMA K dimension)
// is a conflict-free 16-byte LDS read.  Same arithmetic as wn_wgrad_layer.hip (3-product fp16 split, fp32
// accumulation, gradient operand pre-scaled by an exact power of two).
//
// Used for the per-block gradients of widths the one-workgroup-per-block kernel (wn_wgrad_layer.hip:
// R = D = 32 / 64) does not cover -- configs[3], R = D = 128: three jobs per block (two taps of dW_d, dW_r),
// ~5000 workgroups per step -- and (round 2) for head layers of widths 128 / 256 on their own time split: one
// workgroup per (layer, utterance, 1/32 of the utterance) = 256 workgroups per layer computing the WHOLE K x N
// product of their rows, so every operand row is read exactly once.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 wp_h8 __attribute__((ext_vector_type(8)));

// TK_ = row tiles per wave.
template <int LDX, int LDG, int KT, int NT, int LDW = 32 * NT, int TK_ = 2>
__global__ __launch_bounds__(64 * (KT / TK_) * (NT / 2)) void wn_wgrad_pair_kernel(const WnWgPair* jobs, float* ws, float* slab,
                                                                                int64_t P, int B, int T, int spb) {
  constexpr int KC = 32 * KT, NC = 32 * NT, NCH = KC + NC;
  constexpr int PLANE = NCH * 32;                        // bytes of one fp16 plane of a stage
  constexpr int STAGE = 2 * PLANE;
  constexpr int TK = TK_, TN = 2, WKD = KT / TK, NW = (KT / TK) * (NT / TN), NTHR = 64 * NW;
  static_assert(KT % TK == 0 && NT % TN == 0 && 2 * KC <= NTHR && 2 * NC <= NTHR, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 2 * NC * 4];
  float* bpart = reinterpret_cast<float*>(smem + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  const WnWgPair J = jobs[blockIdx.y];
  const int split = blockIdx.x;
  const int ub = split / spb, sp = split % spb;
  int len = (T + spb - 1) / spb;
  len = (len + 15) & ~15;
  const int r0 = sp * len, r1 = min(T, r0 + len);
  const int shift = J.shift;

  float gsc = 1.0f, inv = 1.0f;
  if (J.gmax_off >= 0) {
    const float m = ws[J.gmax_off];
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);
      e = max(-100, min(100, e));
      gsc = ldexpf(1.0f, -e);
      inv = ldexpf(1.0f, e);
    }
  }

  // ---- this thread's two units: one X channel and one G channel, each for one half of the chunk ----
  const bool xunit = tid < 2 * KC, gunit = tid < 2 * NC;
  const int cx = xunit ? tid % KC : 0, hx = xunit ? tid / KC : 0;
  const int cg = gunit ? tid % NC : 0, hg = gunit ? tid / NC : 0;
  const int xshift = shift;
  const float* xptr = ws + J.x_off + (int64_t)ub * T * LDX + cx;
  const float* gptr = ws + J.g_off + (int64_t)ub * T * LDG + cg;
  float bsum = 0.f;

  auto load_chunk = [&](int t0, float (&xv)[8], float (&gv)[8]) {
    const int tx = t0 + 8 * hx - xshift;
    const float* px = xptr + (int64_t)tx * LDX;
    const float* pg = gptr + (int64_t)(t0 + 8 * hg) * LDG;
    if (t0 + 16 <= r1 && t0 - shift >= 0) {              // workgroup-uniform: an interior chunk needs no masks
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[e] = xunit ? px[e * LDX] : 0.f;
        gv[e] = gunit ? pg[e * LDG] : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xv[e] = (xunit && t0 + 8 * hx + e < r1 && tx + e >= 0) ? px[e * LDX] : 0.f;
        gv[e] = (gunit && t0 + 8 * hg + e < r1) ? pg[e * LDG] : 0.f;
      }
    }
  };
  auto store_chunk = [&](int stage, const float (&xv)[8], const float (&gv)[8]) {
    unsigned char* st = smem + stage * STAGE;
    if (xunit) {
      wp_h8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)xv[e];
        hi[e] = hh;
        lo[e] = (_Float16)(xv[e] - (float)hh);
      }
      *reinterpret_cast<wp_h8*>(st + cx * 32 + hx * 16) = hi;
      *reinterpret_cast<wp_h8*>(st + PLANE + cx * 32 + hx * 16) = lo;
    }
    if (gunit) {
      wp_h8 hi, lo;
      float s8 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)(gv[e] * gsc);
        hi[e] = hh;
        lo[e] = (_Float16)__builtin_fmaf(gv[e], gsc, -(float)hh);
        s8 += gv[e];
      }
      bsum += s8;
      *reinterpret_cast<wp_h8*>(st + (KC + cg) * 32 + hg * 16) = hi;
      *reinterpret_cast<wp_h8*>(st + PLANE + (KC + cg) * 32 + hg * 16) = lo;
    }
  };

  // ---- this wave's TK x TN output tiles ----
  const int wk = wave % WKD, wn = wave / WKD;
  f32x16 acc[TK][TN];
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto compute = [&](int stage) {
    const unsigned char* st = smem + stage * STAGE + tl * 32 + h * 16;
    if constexpr (TK <= 2) {
      wp_h8 ah[TK], al[TK];
#pragma unroll
      for (int i = 0; i < TK; ++i) {
        ah[i] = *reinterpret_cast<const wp_h8*>(st + (32 * (wk * TK + i)) * 32);
        al[i] = *reinterpret_cast<const wp_h8*>(st + PLANE + (32 * (wk * TK + i)) * 32);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const wp_h8 bh = *reinterpret_cast<const wp_h8*>(st + (KC + 32 * (wn * TN + j)) * 32);
        const wp_h8 bl = *reinterpret_cast<const wp_h8*>(st + PLANE + (KC + 32 * (wn * TN + j)) * 32);
#pragma unroll
        for (int i = 0; i < TK; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
      }
    } else {
      // more row tiles per wave: the G fragments stay, the X fragments are fetched per row tile (register pressure:
      // 16 * TK * TN accumulators; same products in the same order per output element)
      wp_h8 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const wp_h8*>(st + (KC + 32 * (wn * TN + j)) * 32);
        bl[j] = *reinterpret_cast<const wp_h8*>(st + PLANE + (KC + 32 * (wn * TN + j)) * 32);
      }
#pragma unroll
      for (int i = 0; i < TK; ++i) {
        const wp_h8 ah = *reinterpret_cast<const wp_h8*>(st + (32 * (wk * TK + i)) * 32);
        const wp_h8 al = *reinterpret_cast<const wp_h8*>(st + PLANE + (32 * (wk * TK + i)) * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- pipeline: global loads two chunks ahead (registers), LDS one chunk ahead, one barrier per chunk ----
  float xa[8], ga[8], xb[8], gb[8];
  if (r0 < r1) {
    load_chunk(r0, xa, ga);
    if (r0 + 16 < r1) load_chunk(r0 + 16, xb, gb);
    store_chunk(0, xa, ga);
  }
  __syncthreads();
  for (int t0 = r0; t0 < r1; t0 += 32) {
    if (t0 + 32 < r1) load_chunk(t0 + 32, xa, ga);
    compute(0);
    if (t0 + 16 < r1) store_chunk(1, xb, gb);
    __syncthreads();
    if (t0 + 16 >= r1) break;
    if (t0 + 48 < r1) load_chunk(t0 + 48, xb, gb);
    compute(1);
    if (t0 + 32 < r1) store_chunk(0, xa, ga);
    __syncthreads();
  }

  // ---- partial results -> this split's slab row (laid out like the flat gradient buffer) ----
  float* row = slab + (int64_t)split * P;
#pragma unroll
  for (int i = 0; i < TK; ++i) {
    float* tbase = row + J.w_off + (int64_t)(32 * (wk * TK + i)) * LDW + tl;      // LDW: row pitch of dW (>= the tile's NC)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tbase[wn_drow(r, h) * LDW + 32 * (wn * TN + j)] = acc[i][j][r] * inv;
  }
  if (J.b_off >= 0) {                                    // bias sums: the two halves of a chunk live in different threads
    if (gunit) bpart[hg * NC + cg] = bsum;
    __syncthreads();
    for (int n = tid; n < NC; n += NTHR) row[J.b_off + n] = bpart[n] + bpart[NC + n];
  }
}

// kind 1: X 128 ch x G 256 ch (a tap of dW_d at R = D = 128; a 128 -> 256 head layer); kind 2: 128 x 128 (dW_r);
// kind 3: 256 x 128; kind 5: a 128-column half of 256 x 256 (two jobs) -- head layers of the reference's default
// head [128, 256] -> 256 classes; kind 4 (256 x 256 in one 1024-thread workgroup) is kept for A/B runs only
int wn_wgrad_pair_kind(int K, int N) {
  if (K == 128 && N == 256) return 1;
  if (K == 128 && N == 128) return 2;
  if (K == 256 && N == 128) return 3;
  if (K == 256 && N == 256) return 5;      // as two 256 x 128 column halves (kind 4, one 1024-thread workgroup, ran at 0.3 TB/s)
  return 0;
}

int wn_launch_wgrad_pairs(int kind, const WnWgPair* d_jobs, int njobs, float* ws, float* slab, int64_t P, int B, int T,
                          int splits_per_b, hipStream_t s) {
  if (njobs <= 0) return WN_OK;
  const dim3 grid((unsigned)(B * splits_per_b), (unsigned)njobs);
  switch (kind) {
    case 1: hipLaunchKernelGGL((wn_wgrad_pair_kernel<128, 256, 4, 8>), grid, dim3(512), 0, s, d_jobs, ws, slab, P, B, T, splits_per_b); break;
    case 2: hipLaunchKernelGGL((wn_wgrad_pair_kernel<128, 128, 4, 4>), grid, dim3(256), 0, s, d_jobs, ws, slab, P, B, T, splits_per_b); break;
    case 3: hipLaunchKernelGGL((wn_wgrad_pair_kernel<256, 128, 8, 4>), grid, dim3(512), 0, s, d_jobs, ws, slab, P, B, T, splits_per_b); break;
    case 4: hipLaunchKernelGGL((wn_wgrad_pair_kernel<256, 256, 8, 8>), grid, dim3(1024), 0, s, d_jobs, ws, slab, P, B, T, splits_per_b); break;
    // a 128-column half of a 256 x 256 product: G rows and dW rows keep their pitch of 256
    case 5: hipLaunchKernelGGL((wn_wgrad_pair_kernel<256, 256, 8, 4, 256>), grid, dim3(512), 0, s, d_jobs, ws, slab, P, B, T, splits_per_b); break;
    default: wn_set_error("wgrad_pairs: unknown kind %d", kind); return WN_E_UNSUPPORTED;
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
