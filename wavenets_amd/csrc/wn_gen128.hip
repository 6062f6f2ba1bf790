// One residual block of one queued-generation time step for R = D = 128 (gfx950): rows = utterances (<= 32 per tile).
// Reference: WaveNetLayer.generate, src/layers.py:226-290 -- the step of WaveNetLayer.call on one time sample.
//
// The training-forward kernel of these blocks (wn_layer16s.hip) STREAMS its 320 KiB of weight images through LDS two
// 16 KiB chunks ahead.  With 32 time steps per wave that hides the stream; with one row per utterance it is ten dependent
// L2 round trips per block (25-30 us a block, 0.96 ms a step for 30 blocks = 1.0 k samples/s per utterance).  Here one
// workgroup of 8 waves takes the block; wave w owns output row tile w of the gated conv (32 of the 256 channels of u)
// and requests ALL 16 k-steps of its weight fragments, hi and lo, straight into registers before anything else (128
// VGPRs, one round trip for the whole conv), waves 0..3 also their 8 k-steps of the 1x1.  The operand rows (two taps x
// 128 channels of at most 32 utterances) are staged once in LDS in B-fragment order.  The gate pairs tiles j and j + 4
// through LDS, the 1x1 reads the z tiles back as they stand.
//
// Bit for bit the arithmetic of wn_layer_fwd_s128_kernel (the sliding window's kernel) per output element: accumulators
// start at the bias (+ conditioning bias), k-steps ascending with lo*hi, hi*lo, hi*hi each, the same gate functions, the
// 1x1 the same way from the z tiles, x_out = o + residual in fp32.  tests: queued == sliding window.
#include "wn_stream.h"

using namespace wn_stream;

namespace {

__device__ __forceinline__ h8 ldg_h8(const void* p) { return *(const __attribute__((address_space(1))) h8*)(p); }
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const __attribute__((address_space(1))) f32x4*)(p); }

}  // namespace

// a.xt[0], a.xt[1]: the two taps' rows [B][128]; a.x = residual source when a.res is null (the newest tap); T == 1
__global__ __launch_bounds__(512, 2) void wn_gen_block128_kernel(WnLayerFwdArgs a) {
  constexpr int R = 128, D = 128, NK1 = 16, NK2 = 8;
  // LDS: operand fragments of the conv [16 k-steps][2][64 lanes] x 16 B | gate tiles [4][16][64] floats | z tiles [4][16][64]
  __shared__ __attribute__((aligned(16))) unsigned char smem[NK1 * 2048 + 2 * 4 * 16 * 64 * 4];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);
  float* gs = reinterpret_cast<float*>(smem + NK1 * 2048);
  float* zs = gs + 4 * 16 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * 32 + tl;                 // utterance of this lane's column
  const bool rok = row < a.B;

  // ---- every weight fragment of this wave, requested at once ----
  h8 wd[NK1][2];
  {
    const char* base = reinterpret_cast<const char*>(a.frag_d) + (int64_t)wave * 2048 + lane * 16;   // k-step c: + c * 16384
#pragma unroll
    for (int c = 0; c < NK1; ++c) {
      wd[c][0] = ldg_h8(base + c * 16384);              // hi
      wd[c][1] = ldg_h8(base + c * 16384 + 1024);       // lo
    }
  }
  h8 wr[NK2][2];
  if (wave < 4) {
    const char* base = reinterpret_cast<const char*>(a.frag_r) + (int64_t)wave * 2048 + lane * 16;   // k-step ks: + ks * 8192
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) {
      wr[ks][0] = ldg_h8(base + ks * 8192);
      wr[ks][1] = ldg_h8(base + ks * 8192 + 1024);
    }
  }
  // ---- operand rows -> LDS in B-fragment order: piece (c, q, lane) = channels 16 kk + 8 q + 4 h .. + 3 of row tl, tap c / 8 ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 512 * i;                        // 2048 pieces of 16 bytes
    const int c = e >> 7, q = (e >> 6) & 1, l = e & 63;
    const int r2 = blockIdx.x * 32 + (l & 31);
    const float* src = (c < 8 ? a.xt[0] : a.xt[1]) + (int64_t)(r2 < a.B ? r2 : 0) * R + 16 * (c & 7) + 8 * q + 4 * (l >> 5);
    f32x4 v = ldg4(src);
    if (r2 >= a.B) v = f32x4{0.f, 0.f, 0.f, 0.f};
    xs[e] = v;
  }
  // ---- accumulators start at the bias (+ per-utterance conditioning bias) ----
  f32x16 u;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const f32x4 bv = ldg4(a.bias_d + 32 * wave + 8 * rq + 4 * h);
    u[4 * rq + 0] = bv.x; u[4 * rq + 1] = bv.y; u[4 * rq + 2] = bv.z; u[4 * rq + 3] = bv.w;
  }
  if (a.cb) {
    const float* cbp = a.cb + (int64_t)(rok ? row : 0) * 2 * D + 4 * h + 32 * wave;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      // (wn_layer_fwd_s128_kernel adds the conditioning bias of the tile's utterance: rows of a tile are one utterance
      // there, one utterance per column here)
      const f32x4 cv = ldg4(cbp + 8 * rq);
      u[4 * rq + 0] += cv.x; u[4 * rq + 1] += cv.y; u[4 * rq + 2] += cv.z; u[4 * rq + 3] += cv.w;
    }
  }
  __syncthreads();
  // ---- gated conv: 16 k-steps ----
#pragma unroll
  for (int c = 0; c < NK1; ++c) {
    const f32x4 q0 = xs[(c * 2 + 0) * 64 + lane], q1 = xs[(c * 2 + 1) * 64 + lane];
    h8 bh, bl;
    split8(q0, q1, bh, bl);
    u = mfma16(wd[c][1], bh, u);
    u = mfma16(wd[c][0], bl, u);
    u = mfma16(wd[c][0], bh, u);
  }
  // ---- gate: tiles 4..7 hold the gate channels of tiles 0..3 ----
  if (wave >= 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gs[((wave - 4) * 16 + r) * 64 + lane] = wn_sigmoid_fast(u[r]);
  }
  __syncthreads();
  float wmax = 0.f;
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sg = gs[(wave * 16 + r) * 64 + lane];
      u[r] = wn_tanh_fast(u[r]) * sg;
      zs[(wave * 16 + r) * 64 + lane] = u[r];
    }
    if (a.z_out && rok) {
      float* zp = a.z_out + (int64_t)row * a.ldz + 32 * wave + 4 * h;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 o;
        o.x = u[4 * rq + 0]; o.y = u[4 * rq + 1]; o.z = u[4 * rq + 2]; o.w = u[4 * rq + 3];
        *reinterpret_cast<f32x4*>(zp + 8 * rq) = o;
      }
    }
  }
  // the 1x1's bias and the residual rows are requested before the barrier (the conv's weight registers are free now):
  // they arrive while the z tiles are exchanged
  f32x4 bv4[4], rv4[4];
  if (wave < 4) {
    const float* rp = (a.res ? a.res : a.x) + (int64_t)(rok ? row : 0) * R + 32 * wave + 4 * h;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      bv4[rq] = ldg4(a.bias_r + 32 * wave + 8 * rq + 4 * h);
      rv4[rq] = a.residual ? ldg4(rp + 8 * rq) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  if (wave < 4) {
    // ---- 1x1: 8 k-steps, B operand = the z tiles as they stand ----
    f32x16 o;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 bv = bv4[rq];
      o[4 * rq + 0] = bv.x; o[4 * rq + 1] = bv.y; o[4 * rq + 2] = bv.z; o[4 * rq + 3] = bv.w;
    }
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) {
      const int jz = ks >> 1, r0 = 8 * (ks & 1);
      f32x4 q0, q1;
      q0.x = zs[(jz * 16 + r0 + 0) * 64 + lane]; q0.y = zs[(jz * 16 + r0 + 1) * 64 + lane];
      q0.z = zs[(jz * 16 + r0 + 2) * 64 + lane]; q0.w = zs[(jz * 16 + r0 + 3) * 64 + lane];
      q1.x = zs[(jz * 16 + r0 + 4) * 64 + lane]; q1.y = zs[(jz * 16 + r0 + 5) * 64 + lane];
      q1.z = zs[(jz * 16 + r0 + 6) * 64 + lane]; q1.w = zs[(jz * 16 + r0 + 7) * 64 + lane];
      h8 bh, bl;
      split8(q0, q1, bh, bl);
      o = mfma16(wr[ks][1], bh, o);
      o = mfma16(wr[ks][0], bl, o);
      o = mfma16(wr[ks][0], bh, o);
    }
    // ---- residual, range guard, x_out ----
    if (rok) {
      float* op = a.x_out + (int64_t)row * R + 32 * wave + 4 * h;
      float* pp = a.o_out ? a.o_out + (int64_t)row * R + 32 * wave + 4 * h : nullptr;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 ov;
        ov.x = o[4 * rq + 0]; ov.y = o[4 * rq + 1]; ov.z = o[4 * rq + 2]; ov.w = o[4 * rq + 3];
        if (pp) *reinterpret_cast<f32x4*>(pp + 8 * rq) = ov;
        wmax = wn_absmax_acc(wmax, ov.x, ov.y, ov.z, ov.w);
        if (a.residual) {
          const f32x4 rv = rv4[rq];
          wmax = wn_absmax_acc(wmax, rv.x, rv.y, rv.z, rv.w);
          ov.x += rv.x; ov.y += rv.y; ov.z += rv.z; ov.w += rv.w;
        }
        *reinterpret_cast<f32x4*>(op + 8 * rq) = ov;
      }
    }
    if (a.absmax_out) {
      wmax = wn_wave_absmax_bits(wmax);
      if (lane == 0) wn_absmax_publish_any(a.absmax_out, wmax);
    }
  }
}

// All residual blocks of one queued-generation step in ONE launch (R = D = 128, kernel size 2, depth 1): the loop over
// blocks of the kernel above inside the workgroup.  Block b's output rows stay in LDS as the newest tap of block b + 1 (a
// D-layout tile IS the next conv's B-operand piece of the same lane) and as its residual; only the older tap comes from
// the block's ring.  Per block: one round trip for the weight fragments, 48 + 24 products, three barriers.
__global__ __launch_bounds__(512, 2) void wn_gen_chain128_kernel(WnGen128Args a) {
  constexpr int R = 128, D = 128, NK1 = 16, NK2 = 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NK1 * 2048 + 2 * 4 * 16 * 64 * 4];
  f32x4* xs = reinterpret_cast<f32x4*>(smem);
  float* gs = reinterpret_cast<float*>(smem + NK1 * 2048);
  float* zs = gs + 4 * 16 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * 32 + tl;
  const bool rok = row < a.B;
  float wmax = 0.f;
  // folded skip contraction (a.skip_tiles == 4: 128 columns): waves 4..7 own one column tile each and carry its accumulator
  // through all blocks -- k ascending over (block, k-step) from zero, as wn_gemm_planes16s_kernel runs it
  const bool skip_on = a.skip_w16_off >= 0;
  f32x16 sacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
  for (int b = 0; b < a.nblocks; ++b) {
    const WnGenBlock g = a.blocks[b];
    const float* ring = a.ws + g.ring_off;
    const float* xold = ring + (int64_t)((a.tau - g.dilation) % g.nslots) * a.B * R;
    const float* xnew = ring + (int64_t)(a.tau % g.nslots) * a.B * R;
    h8 wd[NK1][2];
    {
      const char* base = reinterpret_cast<const char*>(a.ws + g.w16d_off) + (int64_t)wave * 2048 + lane * 16;
#pragma unroll
      for (int c = 0; c < NK1; ++c) {
        wd[c][0] = ldg_h8(base + c * 16384);
        wd[c][1] = ldg_h8(base + c * 16384 + 1024);
      }
    }
    h8 wr[NK2][2];                                     // waves 0..3: the 1x1's fragments; 4..7: the skip contraction's
    f32x4 bv4[4];
    if (wave < 4 || skip_on) {
      const char* base = wave < 4 ? reinterpret_cast<const char*>(a.ws + g.w16r_off) + (int64_t)wave * 2048 + lane * 16
                                  : reinterpret_cast<const char*>(a.ws + a.skip_w16_off) + (int64_t)b * (NK2 * 8192) +
                                        (int64_t)(wave - 4) * 2048 + lane * 16;
#pragma unroll
      for (int ks = 0; ks < NK2; ++ks) {
        wr[ks][0] = ldg_h8(base + ks * 8192);
        wr[ks][1] = ldg_h8(base + ks * 8192 + 1024);
      }
    }
    // operand rows: the older tap from the ring (block 0: the newest tap too, written by the input conv's launch)
    const int npieces = b == 0 ? 2048 : 1024;
    for (int e = tid; e < npieces; e += 512) {
      const int c = e >> 7, q = (e >> 6) & 1, l = e & 63;
      const int r2 = blockIdx.x * 32 + (l & 31);
      const int ch0 = 16 * (c & 7) + 8 * q + 4 * (l >> 5);
      const float* src = (c < 8 ? xold : xnew) + (int64_t)(r2 < a.B ? r2 : 0) * R + ch0;
      f32x4 v;
      if (c >= 8 && a.xin) {
        // block 0's newest tap = the input causal conv of this step (src/model.py:84-88,228), computed here with the
        // arithmetic of wn_inconv_fwd_kernel (fma chain over the taps from zero, then + bias) and written to the ring
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int tap = 0; tap < 2; ++tap) {
          const float xv = a.xin[(int64_t)((a.tau - (1 - tap)) % 2) * a.B + (r2 < a.B ? r2 : 0)];
          const f32x4 wv = ldg4(a.causal_w + (int64_t)tap * R + ch0);
          a0 = fmaf(wv.x, xv, a0); a1 = fmaf(wv.y, xv, a1); a2 = fmaf(wv.z, xv, a2); a3 = fmaf(wv.w, xv, a3);
        }
        const f32x4 bv = ldg4(a.causal_b + ch0);
        v = f32x4{a0 + bv.x, a1 + bv.y, a2 + bv.z, a3 + bv.w};
        if (r2 < a.B) *reinterpret_cast<f32x4*>(a.ws + g.ring_off + (int64_t)(a.tau % g.nslots) * a.B * R + (int64_t)r2 * R + ch0) = v;
      } else {
        v = ldg4(src);
      }
      if (r2 >= a.B) v = f32x4{0.f, 0.f, 0.f, 0.f};
      xs[e] = v;
    }
    f32x16 u;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 bv = ldg4(a.params + g.bias_d_off + 32 * wave + 8 * rq + 4 * h);
      u[4 * rq + 0] = bv.x; u[4 * rq + 1] = bv.y; u[4 * rq + 2] = bv.z; u[4 * rq + 3] = bv.w;
    }
    if (g.cb_off >= 0) {
      const float* cbp = a.ws + g.cb_off + (int64_t)(rok ? row : 0) * 2 * D + 4 * h + 32 * wave;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 cv = ldg4(cbp + 8 * rq);
        u[4 * rq + 0] += cv.x; u[4 * rq + 1] += cv.y; u[4 * rq + 2] += cv.z; u[4 * rq + 3] += cv.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NK1; ++c) {
      const f32x4 q0 = xs[(c * 2 + 0) * 64 + lane], q1 = xs[(c * 2 + 1) * 64 + lane];
      h8 bh, bl;
      split8(q0, q1, bh, bl);
      u = mfma16(wd[c][1], bh, u);
      u = mfma16(wd[c][0], bl, u);
      u = mfma16(wd[c][0], bh, u);
    }
    if (wave >= 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gs[((wave - 4) * 16 + r) * 64 + lane] = wn_sigmoid_fast(u[r]);
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sg = gs[(wave * 16 + r) * 64 + lane];
        u[r] = wn_tanh_fast(u[r]) * sg;
        zs[(wave * 16 + r) * 64 + lane] = u[r];
      }
      if (rok) {
        float* zp = a.ws + a.zrow_off + ((int64_t)b * a.B + row) * D + 32 * wave + 4 * h;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 o;
          o.x = u[4 * rq + 0]; o.y = u[4 * rq + 1]; o.z = u[4 * rq + 2]; o.w = u[4 * rq + 3];
          *reinterpret_cast<f32x4*>(zp + 8 * rq) = o;
        }
      }
      // (requested here, when the conv's weight registers are free: in flight across the barrier)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) bv4[rq] = ldg4(a.params + g.bias_r_off + 32 * wave + 8 * rq + 4 * h);
    }
    __syncthreads();
    if (wave < 4) {
      f32x16 o;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = bv4[rq];
        o[4 * rq + 0] = bv.x; o[4 * rq + 1] = bv.y; o[4 * rq + 2] = bv.z; o[4 * rq + 3] = bv.w;
      }
#pragma unroll
      for (int ks = 0; ks < NK2; ++ks) {
        const int jz = ks >> 1, r0 = 8 * (ks & 1);
        f32x4 q0, q1;
        q0.x = zs[(jz * 16 + r0 + 0) * 64 + lane]; q0.y = zs[(jz * 16 + r0 + 1) * 64 + lane];
        q0.z = zs[(jz * 16 + r0 + 2) * 64 + lane]; q0.w = zs[(jz * 16 + r0 + 3) * 64 + lane];
        q1.x = zs[(jz * 16 + r0 + 4) * 64 + lane]; q1.y = zs[(jz * 16 + r0 + 5) * 64 + lane];
        q1.z = zs[(jz * 16 + r0 + 6) * 64 + lane]; q1.w = zs[(jz * 16 + r0 + 7) * 64 + lane];
        h8 bh, bl;
        split8(q0, q1, bh, bl);
        o = mfma16(wr[ks][1], bh, o);
        o = mfma16(wr[ks][0], bl, o);
        o = mfma16(wr[ks][0], bh, o);
      }
      // residual = this lane's own pieces of the newest tap in LDS; x_out goes back into the same pieces (the next block's
      // newest tap) and to the next block's ring slot
      float* dst = nullptr;
      if (b + 1 < a.nblocks) {
        const WnGenBlock gn = a.blocks[b + 1];
        dst = a.ws + gn.ring_off + (int64_t)(a.tau % gn.nslots) * a.B * R;
      } else if (a.hrow_off >= 0) {
        dst = a.ws + a.hrow_off;
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 ov;
        ov.x = o[4 * rq + 0]; ov.y = o[4 * rq + 1]; ov.z = o[4 * rq + 2]; ov.w = o[4 * rq + 3];
        const int piece = ((8 + 2 * wave + (rq >> 1)) * 2 + (rq & 1)) * 64 + lane;
        if (rok) wmax = wn_absmax_acc(wmax, ov.x, ov.y, ov.z, ov.w);
        if (a.residual) {
          const f32x4 rv = xs[piece];
          if (rok) wmax = wn_absmax_acc(wmax, rv.x, rv.y, rv.z, rv.w);
          ov.x += rv.x; ov.y += rv.y; ov.z += rv.z; ov.w += rv.w;
        }
        xs[piece] = rok ? ov : f32x4{0.f, 0.f, 0.f, 0.f};
        if (dst && rok) *reinterpret_cast<f32x4*>(dst + (int64_t)row * R + 32 * wave + 8 * rq + 4 * h) = ov;
      }
    }
    else if (skip_on) {
#pragma unroll
      for (int ks = 0; ks < NK2; ++ks) {
        const int jz = ks >> 1, r0 = 8 * (ks & 1);
        f32x4 q0, q1;
        q0.x = zs[(jz * 16 + r0 + 0) * 64 + lane]; q0.y = zs[(jz * 16 + r0 + 1) * 64 + lane];
        q0.z = zs[(jz * 16 + r0 + 2) * 64 + lane]; q0.w = zs[(jz * 16 + r0 + 3) * 64 + lane];
        q1.x = zs[(jz * 16 + r0 + 4) * 64 + lane]; q1.y = zs[(jz * 16 + r0 + 5) * 64 + lane];
        q1.z = zs[(jz * 16 + r0 + 6) * 64 + lane]; q1.w = zs[(jz * 16 + r0 + 7) * 64 + lane];
        h8 bh, bl;
        split8(q0, q1, bh, bl);
        sacc = mfma16(wr[ks][1], bh, sacc);
        sacc = mfma16(wr[ks][0], bl, sacc);
        sacc = mfma16(wr[ks][0], bh, sacc);
      }
    }
    // (the next iteration's barrier after staging orders these LDS writes before the conv reads them; zs is rewritten
    // two barriers later)
  }
  if (skip_on && wave >= 4) {
    // + bias, activation, range guard: the epilogue of wn_gemm_planes16s_kernel
    const int j = wave - 4;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int n0 = 32 * j + 8 * rq + 4 * h;
      const f32x4 bv = ldg4(a.ws + a.skip_bias_off + n0);
      f32x4 v;
      v.x = wn_act(sacc[4 * rq + 0] + bv.x, a.skip_act); v.y = wn_act(sacc[4 * rq + 1] + bv.y, a.skip_act);
      v.z = wn_act(sacc[4 * rq + 2] + bv.z, a.skip_act); v.w = wn_act(sacc[4 * rq + 3] + bv.w, a.skip_act);
      if (rok) {
        wmax = wn_absmax_acc(wmax, v.x, v.y, v.z, v.w);
        *reinterpret_cast<f32x4*>(a.ws + a.skiprow_off + (int64_t)row * 128 + n0) = v;
      }
    }
  }
  if (a.guard && (wave < 4 || skip_on)) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) wn_absmax_publish_any(a.guard, wmax);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same step as a RELAY over workgroups: one workgroup per (block, utterance tile), all of them resident at once.
//
// wn_gen_chain128_kernel walks the blocks inside ONE workgroup, and every block starts with a cold round trip for its
// 384 KiB of weight fragments through one CU (13 us a block: 0.39 ms a step, 2.5 k samples/s per utterance).  Here
// workgroup b requests the fragments of ITS block the moment the launch starts -- all blocks at once, every CU its own
// fetch path -- stages the older tap (written by an earlier launch) and runs that tap's half of the gated conv (k-steps
// 0..7 come first in the summation order anyway) while the chain is still upstream.  Then it waits for its block input,
// which workgroup b - 1 hands over through memory, finishes the block and hands its output on.  The skip accumulator
// travels the same way between the waves 4..7 of consecutive workgroups (its summation order is block order), off the
// critical path.  Per element the arithmetic is wn_gen_chain128_kernel's, operation for operation.
//
// Hand-off (guide: Guideline 16, form R2 "the data is the flag"): every handed-over fp32 value travels as one naturally
// aligned 8-byte granule {value, epoch} written by ONE relaxed agent-scope (sc1, write-through) atomic store and read by
// relaxed agent-scope atomic loads (sc1: past the reader's L1), re-read until every tag of the wave equals the launch's
// epoch.  No flags, no fences.  Epoch = the step index inside the wn_generate call (>= 1); wn_generate zeroes the granule
// area once per call.  A workgroup only ever waits for the workgroup before it, which has a smaller index and is
// therefore dispatched no later: the chain drains whatever the residency.  Every wait is bounded: a reader that gives up
// records the block in a.tmo and carries on with what it has (wn_generate reports the call as failed afterwards), so
// the launch ends in any case.
typedef __attribute__((address_space(1))) unsigned long long wn_gu64;
#define WN_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define WN_RELAY_SPINS (1u << 21)

namespace {

__device__ __forceinline__ void granule_store(unsigned long long* g, unsigned epoch, float v) {
  __hip_atomic_store((wn_gu64*)g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), WN_RLX_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* g) {
  return __hip_atomic_load((wn_gu64*)g, WN_RLX_AGENT);
}
// Two granules side by side in one 16-byte access ({value, epoch, value, epoch}): each 8-byte half is a granule of its own
// (a 16-byte sc1 access has never been seen to tear an aligned 8-byte half on gfx950; the halves are checked separately).
// Buffer instructions with aux = 16 (sc1): half the requests of 8-byte atomics -- a request costs the issuing wave ~100
// clocks and the sender's stores sit on the chain's critical path.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule2_store(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned epoch, float v0, float v1) {
  const u32x4 q = {__float_as_uint(v0), epoch, __float_as_uint(v1), epoch};
  __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)off, 0, 16);
}
__device__ __forceinline__ u32x4 granule2_load(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
}
__device__ __forceinline__ void relay_give_up(unsigned* tmo, int code, int lane) {
  if (lane == 0) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)tmo, (unsigned)code, WN_RLX_AGENT);
}

}  // namespace

__global__ __launch_bounds__(512, 2) void wn_gen_relay128_kernel(WnGen128Args a) {
  constexpr int R = 128, D = 128, NK1 = 16, NK2 = 8;
  // LDS: B operands of the conv, ALREADY split into fp16 hi | lo, [16 k-steps][hi, lo][64 lanes] x 16 B -- the split is done
  // once by the thread that stages a piece, not by each of the eight waves in front of every product (the operand
  // conversions sat between the LDS read and the products of every k-step of the critical path) | the newest tap in fp32
  // (the residual) [1024 pieces] | z operands, split the same way, [8 k-steps][hi, lo][64]
  __shared__ __attribute__((aligned(16))) unsigned char smem[NK1 * 2048 + 1024 * 16 + NK2 * 2048];
  __shared__ unsigned x_sent;                          // waves 0..3 that have issued their hand-over stores
  h8* const xh = reinterpret_cast<h8*>(smem);
  f32x4* const xr = reinterpret_cast<f32x4*>(smem + NK1 * 2048);
  h8* const zh = reinterpret_cast<h8*>(smem + NK1 * 2048 + 1024 * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  // quad q (0, 1) of k-step ks of lane l: the split8 of wn_stream.h on one half of the operand
  auto put_quad = [&](h8* base, int ks, int q, int l, const f32x4& v) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    h4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const _Float16 hh = (_Float16)f[e];
      hi[e] = hh;
      lo[e] = (_Float16)(f[e] - (float)hh);
    }
    h4* b4 = reinterpret_cast<h4*>(base);
    b4[((ks * 2 + 0) * 64 + l) * 2 + q] = hi;
    b4[((ks * 2 + 1) * 64 + l) * 2 + q] = lo;
  };
  // acc += W^T b over k-steps K0 .. K0 + NK - 1: operands of k-step k + 1 are read before the products of k-step k issue
  auto mac = [&](f32x16& acc, auto w_of, const h8* bop, auto k0_, auto nk_) {
    constexpr int K0 = decltype(k0_)::value, NK = decltype(nk_)::value;
    h8 bh[2], bl[2];
    bh[0] = bop[(K0 * 2 + 0) * 64 + lane];
    bl[0] = bop[(K0 * 2 + 1) * 64 + lane];
    wn_static_for<NK>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (ks + 1 < NK) {
        bh[(ks + 1) & 1] = bop[((K0 + ks + 1) * 2 + 0) * 64 + lane];
        bl[(ks + 1) & 1] = bop[((K0 + ks + 1) * 2 + 1) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
      }
      acc = mfma16(w_of(K0 + ks, 1), bh[ks & 1], acc);
      acc = mfma16(w_of(K0 + ks, 0), bl[ks & 1], acc);
      acc = mfma16(w_of(K0 + ks, 0), bh[ks & 1], acc);
    });
  };
  const int b = (int)blockIdx.x / a.ntiles, tile = (int)blockIdx.x % a.ntiles;   // block-major: predecessors first
  const int row = tile * 32 + tl;
  const bool rok = row < a.B;
  const unsigned epoch = a.epoch;
  const bool skip_on = a.skip_w16_off >= 0;
  const bool last = b + 1 == a.nblocks;
  float wmax = 0.f;
  // knob 24: s_memtime stamps of tile 0, [block][8]: 0 entry, 1 older tap staged, 2 older tap's products done, 3 input arrived (every wave's pieces in LDS),
  // 4 z ready, 5 output handed on (wave 0); 6 skip accumulator arrived, 7 handed on (wave 4)
#define RL_TS(w, k) do { if (a.ts && tile == 0 && tid == 64 * (w)) a.ts[b * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  RL_TS(0, 0);
  // waves 0..3 carry the rows (the chain's critical path) through the 1x1 and share their SIMDs' matrix pipes with the skip
  // waves 4..7, whose accumulator has a block's worth of slack: issue priority to waves 0..3 (as in wn_gen_chain3_kernel)
  if (wave < 4) __builtin_amdgcn_s_setprio(3);
  if (tid == 0) x_sent = 0;                            // (visible after the first barrier; first read far behind it)
  const WnGenBlock g = a.blocks[b];
  // granule areas of this workgroup's INPUT (written by block b - 1) and OUTPUT (read by block b + 1)
  unsigned long long* const gbase = reinterpret_cast<unsigned long long*>(a.ws + a.relay_off);
  auto xg_of = [&](int blk) { return gbase + ((int64_t)blk * a.ntiles + tile) * 8192; };            // [1024 pieces][4]
  auto sg_of = [&](int blk) { return gbase + ((int64_t)(a.nblocks + blk) * a.ntiles + tile) * 8192 ; };   // [4][64][16] (+ pad)

  // ---- every weight fragment of this wave, requested at once ----
  const float* ring = a.ws + g.ring_off;
  const float* xold = ring + (int64_t)((a.tau - g.dilation) % g.nslots) * a.B * R;
  const float* xnew = ring + (int64_t)(a.tau % g.nslots) * a.B * R;
  h8 wd[NK1][2];
  {
    // Wave w owns a MIXED row tile of the gated conv: rows 0..15 = filter channels 16 w .. 16 w + 15, rows 16..31 = the gate
    // channels of the same 16 (the two 16-row halves of image tiles w / 2 and 4 + w / 2), as in wn_gen_chain3_kernel.  A
    // lane's 16 accumulators are then 8 filter values and THEIR 8 gate values: the gate runs in registers, nothing is
    // exchanged through LDS, and every wave ends up with one k-step of z.  Only the lanes' fetch addresses differ from
    // a plain tile: every output element is the same dot product over the same operands in the same order.
    const int src_tile = tl < 16 ? (wave >> 1) : 4 + (wave >> 1);
    const int src_lane = 32 * h + 16 * (wave & 1) + (tl & 15);
    const char* base = reinterpret_cast<const char*>(a.ws + g.w16d_off) + (int64_t)src_tile * 2048 + src_lane * 16;
#pragma unroll
    for (int c = 0; c < NK1; ++c) {
      wd[c][0] = ldg_h8(base + c * 16384);
      wd[c][1] = ldg_h8(base + c * 16384 + 1024);
    }
  }
  h8 wr[NK2][2];                                       // waves 0..3: the 1x1's fragments; 4..7: the skip contraction's
  if (wave < 4 || skip_on) {
    const char* base = wave < 4 ? reinterpret_cast<const char*>(a.ws + g.w16r_off) + (int64_t)wave * 2048 + lane * 16
                                : reinterpret_cast<const char*>(a.ws + a.skip_w16_off) + (int64_t)b * (NK2 * 8192) +
                                      (int64_t)(wave - 4) * 2048 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < NK2; ++ks) {
      wr[ks][0] = ldg_h8(base + ks * 8192);
      wr[ks][1] = ldg_h8(base + ks * 8192 + 1024);
    }
  }
  // ---- operand rows of the older tap (block 0: of both taps) -> LDS in B-fragment order ----
  const int npieces = b == 0 ? 2048 : 1024;
  for (int e = tid; e < npieces; e += 512) {
    const int c = e >> 7, q = (e >> 6) & 1, l = e & 63;
    const int r2 = tile * 32 + (l & 31);
    const int ch0 = 16 * (c & 7) + 8 * q + 4 * (l >> 5);
    const float* src = (c < 8 ? xold : xnew) + (int64_t)(r2 < a.B ? r2 : 0) * R + ch0;
    f32x4 v;
    if (c >= 8 && a.xin) {
      // block 0's newest tap = the input causal conv of this step (see wn_gen_chain128_kernel)
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int tap = 0; tap < 2; ++tap) {
        const float xv = a.xin[(int64_t)((a.tau - (1 - tap)) % 2) * a.B + (r2 < a.B ? r2 : 0)];
        const f32x4 wv = ldg4(a.causal_w + (int64_t)tap * R + ch0);
        a0 = fmaf(wv.x, xv, a0); a1 = fmaf(wv.y, xv, a1); a2 = fmaf(wv.z, xv, a2); a3 = fmaf(wv.w, xv, a3);
      }
      const f32x4 bv = ldg4(a.causal_b + ch0);
      v = f32x4{a0 + bv.x, a1 + bv.y, a2 + bv.z, a3 + bv.w};
      if (r2 < a.B) *reinterpret_cast<f32x4*>(a.ws + g.ring_off + (int64_t)(a.tau % g.nslots) * a.B * R + (int64_t)r2 * R + ch0) = v;
    } else {
      v = ldg4(src);
    }
    if (r2 >= a.B) v = f32x4{0.f, 0.f, 0.f, 0.f};
    put_quad(xh, c, q, l, v);
    if (c >= 8) xr[e - 1024] = v;
  }
  // accumulator quad rq of the mixed tile: channels 16 w + 8 rq + 4 h .. (rq = 0, 1: filter), D + 16 w + 8 (rq - 2) + 4 h .. (gate)
  auto uch = [&](int rq) { return (rq < 2 ? 0 : D - 16) + 16 * wave + 8 * rq + 4 * h; };
  f32x16 u;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const f32x4 bv = ldg4(a.params + g.bias_d_off + uch(rq));
    u[4 * rq + 0] = bv.x; u[4 * rq + 1] = bv.y; u[4 * rq + 2] = bv.z; u[4 * rq + 3] = bv.w;
  }
  if (g.cb_off >= 0) {
    const float* cbp = a.ws + g.cb_off + (int64_t)(rok ? row : 0) * 2 * D;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 cv = ldg4(cbp + uch(rq));
      u[4 * rq + 0] += cv.x; u[4 * rq + 1] += cv.y; u[4 * rq + 2] += cv.z; u[4 * rq + 3] += cv.w;
    }
  }
  // (far down the chain: one lane watches the hand-off INTO the block before this one, so that the whole workgroup
  // only starts polling when its own input is one block away -- 29 workgroups x 512 lanes re-reading their granules for
  // the whole step would load the fabric the chain's own hand-offs travel on)
  if (b >= 2 && tid == 0) {
    const unsigned long long* w = xg_of(b - 1);
    for (unsigned spins = 0; (unsigned)(granule_load(w) >> 32) != epoch; ) {
      if (++spins > WN_RELAY_SPINS) { relay_give_up(a.tmo, b, 0); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  RL_TS(0, 1);
  // ---- gated conv, older tap: k-steps 0..7 ----
  auto wd_of = [&](int c, int hl) -> const h8& { return wd[c][hl]; };
  mac(u, wd_of, xh, std::integral_constant<int, 0>{}, std::integral_constant<int, NK1 / 2>{});
  RL_TS(0, 2);
  // ---- the block input of this step, from block b - 1: pieces 1024..2047 (2 per thread, 4 granules each) ----
  if (b > 0) {
    const unsigned long long* xg = xg_of(b);
    f32x4 v[2];
    bool lv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i;
      lv[i] = tile * 32 + (e & 31) < a.B;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(xg), 0, 65536, 0x00020000);
    for (unsigned spins = 0;;) {
      bool ok = true;
      u32x4 q[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (lv[i]) {
          q[i][0] = granule2_load(xrs, (unsigned)(tid + 512 * i) * 32u);
          q[i][1] = granule2_load(xrs, (unsigned)(tid + 512 * i) * 32u + 16u);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (lv[i]) {
          ok = ok && q[i][0].y == epoch && q[i][0].w == epoch && q[i][1].y == epoch && q[i][1].w == epoch;
          v[i] = f32x4{__uint_as_float(q[i][0].x), __uint_as_float(q[i][0].z), __uint_as_float(q[i][1].x), __uint_as_float(q[i][1].z)};
        }
      }
      if (__all(ok)) break;                            // wave-uniform
      if (++spins > WN_RELAY_SPINS) { relay_give_up(a.tmo, 1000 + b, lane); break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i;                       // piece 1024 + e: k-step 8 + (e >> 7), quad (e >> 6) & 1, lane e & 63
      put_quad(xh, 8 + (e >> 7), (e >> 6) & 1, e & 63, v[i]);
      xr[e] = v[i];
    }
    __syncthreads();
    RL_TS(0, 3);
  }
  // ---- gated conv, newest tap: k-steps 8..15 ----
  mac(u, wd_of, xh, std::integral_constant<int, NK1 / 2>{}, std::integral_constant<int, NK1 / 2>{});
  // The skip accumulator of block b - 1 was handed over BEFORE that block's rows (its products run while the 1x1 of the
  // rows is still busy), so it has landed by now: waves 4..7 request it here, under the gate's transcendentals, and check the
  // tags once z is in LDS.  Kept out of the phase in which waves 0..3 issue their hand-over stores: those are the chain's
  // critical path, they share the CU's one memory queue with every other request of the workgroup, and a wave that
  // polls 8 x 16 B per lane there was measured to stretch "z ready -> rows handed on" from 1.8 to 2.9 us.
  const bool skip_recv = skip_on && wave >= 4 && b > 0 && rok;
  const int sj = wave - 4;
  u32x4 sq[8];
  if (skip_recv) {
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(sg_of(b), 0, 65536, 0x00020000);
#pragma unroll
    for (int r = 0; r < 8; ++r) sq[r] = granule2_load(srs, (unsigned)(sj * 64 + lane) * 128u + 16u * r);
  }
  // the gate, in registers: accumulators 0..7 are filter channels, 8..15 their gate channels; z = k-step `wave` of the
  // 1x1 and the skip contraction, split into fp16 hi | lo on the way into LDS
  f32x4 zq[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    zq[q].x = wn_tanh_fast(u[4 * q + 0]) * wn_sigmoid_fast(u[8 + 4 * q + 0]);
    zq[q].y = wn_tanh_fast(u[4 * q + 1]) * wn_sigmoid_fast(u[8 + 4 * q + 1]);
    zq[q].z = wn_tanh_fast(u[4 * q + 2]) * wn_sigmoid_fast(u[8 + 4 * q + 2]);
    zq[q].w = wn_tanh_fast(u[4 * q + 3]) * wn_sigmoid_fast(u[8 + 4 * q + 3]);
    put_quad(zh, wave, q, lane, zq[q]);
  }
  f32x4 bv4[4];
  if (wave < 4) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) bv4[rq] = ldg4(a.params + g.bias_r_off + 32 * wave + 8 * rq + 4 * h);
  }
  __syncthreads();
  RL_TS(0, 4);
  auto store_zrow = [&]() {                              // gated activations of this block (later launches only)
    if (rok) {
      float* zp = a.ws + a.zrow_off + ((int64_t)b * a.B + row) * D + 16 * wave + 4 * h;
      *reinterpret_cast<f32x4*>(zp) = zq[0];
      *reinterpret_cast<f32x4*>(zp + 8) = zq[1];
    }
  };
  if (wave < 4) {
    f32x16 o;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 bv = bv4[rq];
      o[4 * rq + 0] = bv.x; o[4 * rq + 1] = bv.y; o[4 * rq + 2] = bv.z; o[4 * rq + 3] = bv.w;
    }
    auto wr_of = [&](int ks, int hl) -> const h8& { return wr[ks][hl]; };
    mac(o, wr_of, zh, std::integral_constant<int, 0>{}, std::integral_constant<int, NK2>{});
    // residual = this lane's own pieces of the newest tap; the sum goes to block b + 1 as granules FIRST (its workgroup
    // is waiting for them), then to that block's ring slot of this time step (read by later launches only)
    unsigned long long* xgn = last ? nullptr : xg_of(b + 1);
    float* dst = nullptr;
    if (!last) {
      const WnGenBlock gn = a.blocks[b + 1];
      dst = a.ws + gn.ring_off + (int64_t)(a.tau % gn.nslots) * a.B * R;
    } else if (a.hrow_off >= 0) {
      dst = a.ws + a.hrow_off;
    }
    const __amdgpu_buffer_rsrc_t xnrs = __builtin_amdgcn_make_buffer_rsrc(xgn ? xgn : gbase, 0, 65536, 0x00020000);
    f32x4 ov[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      ov[rq].x = o[4 * rq + 0]; ov[rq].y = o[4 * rq + 1]; ov[rq].z = o[4 * rq + 2]; ov[rq].w = o[4 * rq + 3];
      const int piece = ((8 + 2 * wave + (rq >> 1)) * 2 + (rq & 1)) * 64 + lane;
      if (rok) wmax = wn_absmax_acc(wmax, ov[rq].x, ov[rq].y, ov[rq].z, ov[rq].w);
      if (a.residual) {
        const f32x4 rv = xr[piece - 1024];
        if (rok) wmax = wn_absmax_acc(wmax, rv.x, rv.y, rv.z, rv.w);
        ov[rq].x += rv.x; ov[rq].y += rv.y; ov[rq].z += rv.z; ov[rq].w += rv.w;
      }
      if (xgn && rok && b != a.mute_block) {
        granule2_store(xnrs, (unsigned)(piece - 1024) * 32u, epoch, ov[rq].x, ov[rq].y);
        granule2_store(xnrs, (unsigned)(piece - 1024) * 32u + 16u, epoch, ov[rq].z, ov[rq].w);
      }
    }
    if (lane == 0) __hip_atomic_fetch_add(&x_sent, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    RL_TS(0, 5);
    if (dst && rok) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) *reinterpret_cast<f32x4*>(dst + (int64_t)row * R + 32 * wave + 8 * rq + 4 * h) = ov[rq];
    }
    store_zrow();
  }
  if (wave >= 4 && skip_on) {
    // ---- folded skip contraction: the accumulator of column tile wave - 4 arrives from block b - 1 (same lane, same
    //      register), takes this block's 24 products, and travels on (or, after the last block, through the epilogue of
    //      wn_gemm_planes16s_kernel) ----
    const int j = wave - 4;
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    if (skip_recv) {
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(sg_of(b), 0, 65536, 0x00020000);
      const unsigned soff = (unsigned)(j * 64 + lane) * 128u;
      for (unsigned spins = 0;;) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          ok = ok && sq[r].y == epoch && sq[r].w == epoch;
          sacc[2 * r] = __uint_as_float(sq[r].x);
          sacc[2 * r + 1] = __uint_as_float(sq[r].z);
        }
        if (ok) break;                                  // (per lane: the lanes of a wave leave the loop one by one)
        if (++spins > WN_RELAY_SPINS) { relay_give_up(a.tmo, 2000 + b, 0); break; }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int r = 0; r < 8; ++r) sq[r] = granule2_load(srs, soff + 16u * r);
      }
    }
    RL_TS(4, 6);
    auto wr_of = [&](int ks, int hl) -> const h8& { return wr[ks][hl]; };
    mac(sacc, wr_of, zh, std::integral_constant<int, 0>{}, std::integral_constant<int, NK2>{});
    if (!last) {
      // (behind the rows' hand-over stores, see above; bounded like every wait of this kernel)
      for (unsigned spins = 0; __hip_atomic_load(&x_sent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u && spins < WN_RELAY_SPINS; ++spins)
        __builtin_amdgcn_s_sleep(1);
      if (rok) {
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(sg_of(b + 1), 0, 65536, 0x00020000);
#pragma unroll
        for (int r = 0; r < 8; ++r) granule2_store(srs, (unsigned)(j * 64 + lane) * 128u + 16u * r, epoch, sacc[2 * r], sacc[2 * r + 1]);
      }
      RL_TS(4, 7);
    } else {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n0 = 32 * j + 8 * rq + 4 * h;
        const f32x4 bv = ldg4(a.ws + a.skip_bias_off + n0);
        f32x4 v;
        v.x = wn_act(sacc[4 * rq + 0] + bv.x, a.skip_act); v.y = wn_act(sacc[4 * rq + 1] + bv.y, a.skip_act);
        v.z = wn_act(sacc[4 * rq + 2] + bv.z, a.skip_act); v.w = wn_act(sacc[4 * rq + 3] + bv.w, a.skip_act);
        if (rok) {
          wmax = wn_absmax_acc(wmax, v.x, v.y, v.z, v.w);
          *reinterpret_cast<f32x4*>(a.ws + a.skiprow_off + (int64_t)row * 128 + n0) = v;
        }
      }
    }
  }
  if (wave >= 4) store_zrow();                          // (last: nothing of waves 4..7 may sit in the memory queue before the rows' hand-over)
  if (a.guard && (wave < 4 || (skip_on && last))) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) wn_absmax_publish_any(a.guard, wmax);
  }
}

// floats of workspace the relay's granule areas take: [x | skip accumulator][block][tile][8192 granules]
int64_t wn_gen_relay128_floats(int B, int nblocks) { return (int64_t)2 * nblocks * ((B + 31) / 32) * 8192 * 2; }

// knob 24: phase stamps of the relay (profiling hook, read by wn_debug_relay_ts)
static unsigned long long* g_relay_ts = nullptr;
extern "C" int wn_debug_relay_ts(unsigned long long* out, int nblocks) {
  if (!g_relay_ts || nblocks > 128) return -1;
  return (int)hipMemcpy(out, g_relay_ts, (size_t)nblocks * 8 * 8, hipMemcpyDeviceToHost);
}

int wn_launch_gen_relay128(const WnGen128Args& a0, hipStream_t s) {
  WnGen128Args a = a0;
  a.ts = nullptr;
  a.mute_block = wn_debug_get(3) - 1;                  // knob 3 = b + 1: block b withholds its rows (watchdog test)
  if (wn_debug_get(24) && a.nblocks <= 128) {
    if (!g_relay_ts) { (void)hipMalloc((void**)&g_relay_ts, 128 * 8 * 8); (void)hipMemset(g_relay_ts, 0, 128 * 8 * 8); }
    a.ts = g_relay_ts;
  }
  if (a.B <= 0 || a.nblocks <= 0) return WN_OK;
  if (a.ntiles != (a.B + 31) / 32 || a.epoch == 0 || !a.tmo || a.relay_off < 0) { wn_set_error("gen_relay128: bad arguments"); return WN_E_INVALID; }
  hipLaunchKernelGGL(wn_gen_relay128_kernel, dim3((unsigned)(a.ntiles * a.nblocks)), dim3(512), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int wn_launch_gen_chain128(const WnGen128Args& a, hipStream_t s) {
  if (a.B <= 0 || a.nblocks <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_gen_chain128_kernel, dim3((unsigned)((a.B + 31) / 32)), dim3(512), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int wn_gen_block128_supported(int R, int D, int KS) { return R == 128 && D == 128 && KS == 2; }

int wn_launch_gen_block128(const WnLayerFwdArgs& a, hipStream_t s) {
  if (!wn_gen_block128_supported(a.R, a.D, a.KS) || a.T != 1 || !a.xt[0] || !a.xt[1] || a.ag_out || (a.z_out && a.ldz % 4 != 0)) {
    wn_set_error("gen_block128: unsupported call (R=%d D=%d KS=%d T=%d)", a.R, a.D, a.KS, a.T);
    return WN_E_UNSUPPORTED;
  }
  if (a.B <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_gen_block128_kernel, dim3((unsigned)((a.B + 31) / 32)), dim3(512), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
