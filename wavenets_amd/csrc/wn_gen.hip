// One generation time step through the input conv and ALL residual blocks in a single launch (gfx950).
// Reference semantics: WaveNet.generate / _generation (src/model.py:241-307) with the per-layer queues
// the reference left as a TODO (README.md:16, src/layers.py:226-290).
//
// Rows are utterances (lane & 31 = utterance, 32 per wave); a wave carries its utterances through the
// whole block chain in registers: the output tile of block b is, unchanged, the current-tap B operand
// of block b + 1 (wn_common.h), the older taps come from per-block ring buffers in HBM, the gated
// activations z go to the row buffer that feeds the folded skip contraction.  The arithmetic is the
// split-precision MFMA sequence of wn_layer16.hip in the same order, so every value is bit-identical to
// what the sliding-window path computes for the same sample.  Weights (fp16 hi|lo images) stream from
// L2: 2.4 MB per step at configs[1].
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 gn_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void gn_split8(const f32x4& q0, const f32x4& q1, gn_h8& hi, gn_h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)v[e];
    hi[e] = h;
    lo[e] = (_Float16)(v[e] - (float)h);
  }
}

template <int R32, int D32, int KS>
__global__ __launch_bounds__(64) void wn_gen_blocks_kernel(WnGenStepArgs a) {
  constexpr int R = 32 * R32, D = 32 * D32, JU = 2 * D32, QR = R / 8;
  constexpr int KS1 = KS * R / 16, KS2 = D / 16;
  const int lane = threadIdx.x & 63;
  const int tl = lane & 31, h = lane >> 5;
  const int utt = blockIdx.x * 32 + tl;
  const bool live = utt < a.B;
  const int ur = live ? utt : 0;                     // clamped row for loads

  // ---- input causal conv (C_in = 1): the k-ordered fma chain of the fp32 MFMA path, then + bias ----
  f32x16 xc[R32];
  {
    float xs[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) xs[t] = a.xin[(int64_t)((a.tau - (KS - 1 - t)) % KS) * a.B + ur];
#pragma unroll
    for (int j = 0; j < R32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * j + wn_drow(r, h);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) acc = fmaf(a.causal_w[t * R + c], xs[t], acc);
        xc[j][r] = acc + a.causal_b[c];
      }
  }

  for (int b = 0; b < a.nblocks; ++b) {
    const WnGenBlock blk = a.blocks[b];
    float* ring = a.ws + blk.ring_off;
    // ---- this block's input at time tau goes into its ring (read again d, 2d, ... steps later) ----
    if (live) {
      float* dst = ring + ((int64_t)(a.tau % blk.nslots) * a.B + utt) * R;
#pragma unroll
      for (int j = 0; j < R32; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 o;
          o.x = xc[j][4 * rq + 0]; o.y = xc[j][4 * rq + 1]; o.z = xc[j][4 * rq + 2]; o.w = xc[j][4 * rq + 3];
          *reinterpret_cast<f32x4*>(dst + 32 * j + 8 * rq + 4 * h) = o;
        }
    }
    // ---- taps: older ones from the ring, the newest from registers ----
    f32x4 xq[KS][QR];
#pragma unroll
    for (int t = 0; t + 1 < KS; ++t) {
      const int64_t slot = (a.tau - (int64_t)(KS - 1 - t) * blk.dilation) % blk.nslots;
      const float* src = ring + (slot * a.B + ur) * R + 4 * h;
#pragma unroll
      for (int q = 0; q < QR; ++q) xq[t][q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
    }
#pragma unroll
    for (int q = 0; q < QR; ++q) {
      const int j = q / 4, rq = q % 4;
      xq[KS - 1][q] = f32x4{xc[j][4 * rq + 0], xc[j][4 * rq + 1], xc[j][4 * rq + 2], xc[j][4 * rq + 3]};
    }
    // ---- u = b_d (+ cb) + sum_tap W_tap^T x_tap ----
    f32x16 u[JU];
    const float* bias_d = a.params + blk.bias_d_off;
#pragma unroll
    for (int j = 0; j < JU; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_d + 32 * j + 8 * rq + 4 * h);
        u[j][4 * rq + 0] = bv.x; u[j][4 * rq + 1] = bv.y; u[j][4 * rq + 2] = bv.z; u[j][4 * rq + 3] = bv.w;
      }
    if (blk.cb_off >= 0) {
      const float* cbp = a.ws + blk.cb_off + (int64_t)ur * 2 * D + 4 * h;
#pragma unroll
      for (int j = 0; j < JU; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(cbp + 32 * j + 8 * rq);
          u[j][4 * rq + 0] += cv.x; u[j][4 * rq + 1] += cv.y; u[j][4 * rq + 2] += cv.z; u[j][4 * rq + 3] += cv.w;
        }
    }
    // a single wave per 32 utterances has nobody to hide L2 latency behind: the weight fragments of
    // k-step ks + PD are requested before the MFMAs of k-step ks (register ring, schedule pinned)
    const gn_h8* wd = reinterpret_cast<const gn_h8*>(a.ws + blk.w16d_off) + lane;
    const gn_h8* wr = reinterpret_cast<const gn_h8*>(a.ws + blk.w16r_off) + lane;
    constexpr int PD = 2;
    gn_h8 wring[PD + 1][JU][2];
    wn_static_for<PD>([&](auto sc) {
      constexpr int ks = decltype(sc)::value;
      if constexpr (ks < KS1) {
#pragma unroll
        for (int j = 0; j < JU; ++j) {
          wring[ks % (PD + 1)][j][0] = wd[((ks * JU + j) * 2 + 0) * 64];
          wring[ks % (PD + 1)][j][1] = wd[((ks * JU + j) * 2 + 1) * 64];
        }
      }
    });
    // conv1 fragments (all of them: KS2 * R32 * 2 <= 16 vectors) are requested up front as well
    gn_h8 rfr[KS2][R32][2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
      for (int j = 0; j < R32; ++j) {
        rfr[ks][j][0] = wr[((ks * R32 + j) * 2 + 0) * 64];
        rfr[ks][j][1] = wr[((ks * R32 + j) * 2 + 1) * 64];
      }
    wn_static_for<KS1>([&](auto sc) {
      constexpr int ks = decltype(sc)::value;
      constexpr int tap = ks / (R / 16), kk = ks % (R / 16);
      if constexpr (ks + PD < KS1) {
#pragma unroll
        for (int j = 0; j < JU; ++j) {
          wring[(ks + PD) % (PD + 1)][j][0] = wd[(((ks + PD) * JU + j) * 2 + 0) * 64];
          wring[(ks + PD) % (PD + 1)][j][1] = wd[(((ks + PD) * JU + j) * 2 + 1) * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      gn_h8 bh, bl;
      gn_split8(xq[tap][2 * kk], xq[tap][2 * kk + 1], bh, bl);
#pragma unroll
      for (int j = 0; j < JU; ++j) {
        const gn_h8 ah = wring[ks % (PD + 1)][j][0];
        const gn_h8 al = wring[ks % (PD + 1)][j][1];
        u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, u[j], 0, 0, 0);
        u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, u[j], 0, 0, 0);
        u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, u[j], 0, 0, 0);
      }
    });
    // ---- gate, z row for the folded skip contraction ----
#pragma unroll
    for (int j = 0; j < D32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) u[j][r] = wn_tanh_fast(u[j][r]) * wn_sigmoid_fast(u[j + D32][r]);
    if (live) {
      float* zdst = a.ws + a.zrow_off + ((int64_t)b * a.B + utt) * D;
#pragma unroll
      for (int j = 0; j < D32; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 o;
          o.x = u[j][4 * rq + 0]; o.y = u[j][4 * rq + 1]; o.z = u[j][4 * rq + 2]; o.w = u[j][4 * rq + 3];
          *reinterpret_cast<f32x4*>(zdst + 32 * j + 8 * rq + 4 * h) = o;
        }
    }
    // ---- o = b_r + W_r^T z ; x_next = o (+ x) ----
    f32x16 o[R32];
    const float* bias_r = a.params + blk.bias_r_off;
#pragma unroll
    for (int j = 0; j < R32; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_r + 32 * j + 8 * rq + 4 * h);
        o[j][4 * rq + 0] = bv.x; o[j][4 * rq + 1] = bv.y; o[j][4 * rq + 2] = bv.z; o[j][4 * rq + 3] = bv.w;
      }
    wn_static_for<KS2>([&](auto sc) {
      constexpr int ks = decltype(sc)::value;
      constexpr int jz = ks / 2, r0 = 8 * (ks % 2);
      const f32x4 q0 = {u[jz][r0 + 0], u[jz][r0 + 1], u[jz][r0 + 2], u[jz][r0 + 3]};
      const f32x4 q1 = {u[jz][r0 + 4], u[jz][r0 + 5], u[jz][r0 + 6], u[jz][r0 + 7]};
      gn_h8 bh, bl;
      gn_split8(q0, q1, bh, bl);
#pragma unroll
      for (int j = 0; j < R32; ++j) {
        const gn_h8 ah = rfr[ks][j][0];
        const gn_h8 al = rfr[ks][j][1];
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, o[j], 0, 0, 0);
      }
    });
#pragma unroll
    for (int j = 0; j < R32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) xc[j][r] = a.residual ? o[j][r] + xc[j][r] : o[j][r];
  }
  // ---- the last block output feeds the head when use_skip is False ----
  if (a.hrow_off >= 0 && live) {
    float* dst = a.ws + a.hrow_off + (int64_t)utt * R;
#pragma unroll
    for (int j = 0; j < R32; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 o;
        o.x = xc[j][4 * rq + 0]; o.y = xc[j][4 * rq + 1]; o.z = xc[j][4 * rq + 2]; o.w = xc[j][4 * rq + 3];
        *reinterpret_cast<f32x4*>(dst + 32 * j + 8 * rq + 4 * h) = o;
      }
  }
}

int wn_gen_blocks_supported(int R, int D, int KS) {
  if (R == 32 && D == 32) return KS == 2 || KS == 3;
  if (R == 64 && D == 64) return KS == 2;
  return 0;
}

int wn_launch_gen_blocks(const WnGenStepArgs& a, int R, int KS, hipStream_t s) {
  const unsigned gx = (unsigned)((a.B + 31) / 32);
  if (R == 32 && KS == 2) hipLaunchKernelGGL((wn_gen_blocks_kernel<1, 1, 2>), dim3(gx), dim3(64), 0, s, a);
  else if (R == 32 && KS == 3) hipLaunchKernelGGL((wn_gen_blocks_kernel<1, 1, 3>), dim3(gx), dim3(64), 0, s, a);
  else if (R == 64 && KS == 2) hipLaunchKernelGGL((wn_gen_blocks_kernel<2, 2, 2>), dim3(gx), dim3(64), 0, s, a);
  else { wn_set_error("gen_blocks: unsupported shape"); return WN_E_UNSUPPORTED; }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
