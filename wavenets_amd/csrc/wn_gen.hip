// One generation time step through the input conv and ALL residual blocks in a single launch (gfx950).
// Reference semantics: WaveNet.generate / _generation (src/model.py:241-307) with the per-layer queues
// the reference left as a TODO (README.md:16, src/layers.py:226-290).
//
// Rows are utterances (lane & 31 = utterance, 32 per wave).  Two launches per step:
//
//  wn_gen_pre_kernel    grid (utterance tiles, blocks).  Everything of a block's gated conv that does
//                       NOT depend on this step's chain: u0_b = b_d (+ cb) + sum_{tap < k-1} W_tap^T x_b[tau - ..]
//                       (the older taps sit in the block's ring buffer since earlier steps).  All blocks
//                       in parallel; the accumulators go to a lane-major scratch image.
//  wn_gen_chain_kernel  grid (utterance tiles).  One wave carries its utterances through the block chain
//                       in registers: u = u0_b + W_{k-1}^T x_b[tau], gate, z row out, 1x1, residual; the
//                       output tile of block b is, unchanged, the B operand of block b + 1 (wn_common.h).
//                       While block b computes, the wave's own LDS-DMA brings block b + 1's weight
//                       fragments and u0 image into the other half of LDS (no registers, no barrier:
//                       a single wave only waits on its own vmcnt).
//
// The per-element MFMA sequence (bias, taps in k order, lo*hi, hi*lo, hi*hi) is the one of
// wn_layer16.hip, merely cut between two launches, so every value is bit-identical to what the
// sliding-window path computes for the same sample.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"
#include "wn_sample.h"

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
struct GnShape {
  static constexpr int R = 32 * R32, D = 32 * D32, JU = 2 * D32, QR = R / 8;
  static constexpr int KSR = R / 16;                  // k-steps per tap
  static constexpr int KS0 = (KS - 1) * KSR;          // k-steps of the older taps (pre kernel)
  static constexpr int KS2 = D / 16;
  static constexpr int WD_BYTES = KSR * JU * 2048;    // newest tap's hi|lo fragments
  static constexpr int WR_BYTES = KS2 * R32 * 2048;   // conv1
  static constexpr int U0_BYTES = JU * 4096;          // JU tiles x 16 accumulators x 64 lanes x 4 B
  static constexpr int BUF_BYTES = WD_BYTES + WR_BYTES + U0_BYTES;
};

template <int R32, int D32, int KS>
__device__ __forceinline__ void gn_pre_body(const WnGenStepArgs& a, int tile, int b, int ntiles) {
  using S = GnShape<R32, D32, KS>;
  constexpr int R = S::R, D = S::D, JU = S::JU, QR = S::QR, KSR = S::KSR, KS0 = S::KS0;
  const int lane = threadIdx.x & 63;
  const int tl = lane & 31, h = lane >> 5;
  const int utt = tile * 32 + tl;
  const int ur = utt < a.B ? utt : 0;
  const WnGenBlock blk = a.blocks[b];
  // weight fragments of the older taps: all requested at once (one L2 round trip)
  const gn_h8* wd = reinterpret_cast<const gn_h8*>(a.ws + blk.w16d_off) + lane;
  gn_h8 wf[KS0][JU][2];
#pragma unroll
  for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
    for (int j = 0; j < JU; ++j) {
      wf[ks][j][0] = wd[((ks * JU + j) * 2 + 0) * 64];
      wf[ks][j][1] = wd[((ks * JU + j) * 2 + 1) * 64];
    }
  const float* ring = a.ws + blk.ring_off;
  f32x4 xq[KS - 1][QR];
#pragma unroll
  for (int t = 0; t + 1 < KS; ++t) {
    // 32-bit index arithmetic (the host checks nslots * B * R < 2^31): a 64-bit modulo is ~100 instructions
    const int slot = (int)((unsigned)((int)a.tau - (KS - 1 - t) * blk.dilation) % (unsigned)blk.nslots);
    const float* src = ring + (slot * a.B + ur) * R + 4 * h;
#pragma unroll
    for (int q = 0; q < QR; ++q) xq[t][q] = *reinterpret_cast<const f32x4*>(src + 8 * q);
  }
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
  wn_static_for<KS0>([&](auto sc) {
    constexpr int ks = decltype(sc)::value;
    constexpr int tap = ks / KSR, kk = ks % KSR;
    gn_h8 bh, bl;
    gn_split8(xq[tap][2 * kk], xq[tap][2 * kk + 1], bh, bl);
#pragma unroll
    for (int j = 0; j < JU; ++j) {
      u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j][1], bh, u[j], 0, 0, 0);
      u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j][0], bl, u[j], 0, 0, 0);
      u[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j][0], bh, u[j], 0, 0, 0);
    }
  });
  // lane-major image [b][tile][JU * 4 quads][64 lanes] of float4: contiguous KiB per quad (LDS-DMA friendly)
  f32x4* dst = reinterpret_cast<f32x4*>(a.ws + a.u0_off) + ((int64_t)b * ntiles + tile) * (JU * 4) * 64 + lane;
#pragma unroll
  for (int j = 0; j < JU; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      dst[(j * 4 + rq) * 64] = f32x4{u[j][4 * rq + 0], u[j][4 * rq + 1], u[j][4 * rq + 2], u[j][4 * rq + 3]};
}
template <int R32, int D32, int KS>
__global__ __launch_bounds__(64) void wn_gen_pre_kernel(WnGenStepArgs a) {
  gn_pre_body<R32, D32, KS>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// workgroup barrier that only drains this wave's LDS/SMEM traffic: __syncthreads() would also wait for
// the global prefetches and LDS-DMA that are meant to stay in flight across phases
#define GN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")


// Chain kernel, second form: weights travel in REGISTERS, several blocks ahead.
// A kernel starts with a cold L2 (the per-XCD L2s are invalidated at kernel boundaries), so every weight line of a
// generation step comes from the memory side, ~2 us away -- about two blocks' worth of arithmetic.  One workgroup per 32
// utterances, at most 10 waves (168 registers a lane), three exclusive roles, each with its own copy of the block loop
// (same barrier count: two a block, (2) z operands visible, (3) x operands visible) so that its registers are allocated
// apart from the others':
//   chain wave w < JU      phase A: a MIXED u tile (16 filter channels and their 16 gate channels) = u0 + W_{k-1}^T x, then
//                          the gate in registers: z k-step w.  Holds the newest-tap fragments and the u0 image of ITS
//                          tile for blocks b, b + 1.  (Plain tiles would need the u tiles exchanged through LDS and a
//                          barrier between phase A and the gate: 0.069 -> 0.061 ms per step without.)
//   conv1 wave j < R / 32  carries the block input x; phase C: o tile j = b_r + W_r^T z, x_next = o (+ x).  Holds the
//                          conv1 fragments and bias of its tile for blocks b, b + 1.
//   skip wave s            two column tiles of the folded skip contraction, acc += W_{s,b}^T z (k order = block order):
//                          tile 2s right after z_b is visible, tile 2s+1 one phase later (z operands are double-buffered)
//                          while the chain waves are in their transcendental phase and the MFMA pipe is idle.
// A workgroup's vector memory operations go through one 64 B / clock address pipe -- 128 KB of weights a block is ~2000
// clocks of it, against ~2600 of arithmetic -- and a wave that issues into a full pipe stalls.  So no wave issues its
// refill between its arithmetic and the barrier that ends its phase: the chain waves issue theirs in phase C (when they
// idle), the conv1 waves in phase A, the skip waves in phases A and B.  LDS only holds the exchange buffers (x operands,
// 2 x z operands) and a copy of the block table.
#define WN_GEN_CHAIN_MAX_BLOCKS 128
// acc += W^T b over NK k-steps (hi|lo B operands in LDS, 2 KB a k-step), the three products of a k-step in the order of
// the training kernels.  The operands of k-step k + 1 are requested BEFORE the products of k-step k are issued: written
// the plain way the compiler reads, waits, multiplies, reads, ... and every k-step pays an LDS round trip (measured on
// the conv1 waves: 1220 -> 550 cycles for the 12 products).
template <int NK>
__device__ __forceinline__ void gn_mac(f32x16& acc, const gn_h8 (&w)[NK][2], const gn_h8* zl) {
  gn_h8 bh[2], bl[2];
  bh[0] = zl[0];
  bl[0] = zl[64];
  wn_static_for<NK>([&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    if constexpr (ks + 1 < NK) {
      bh[(ks + 1) & 1] = zl[((ks + 1) * 2 + 0) * 64];
      bl[(ks + 1) & 1] = zl[((ks + 1) * 2 + 1) * 64];
      __builtin_amdgcn_sched_barrier(0);            // keep the two reads above the products below
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][1], bh[ks & 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][0], bl[ks & 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks][0], bh[ks & 1], acc, 0, 0, 0);
  });
}
#define WN_GEN_HELPERS_PER_XCD 2
// The prefetches are issued as inline asm and waited for by hand: the compiler's own s_waitcnt placement drains vmcnt to 0
// at every use inside a loop, which would cut the distance of a 3-blocks-ahead fetch to one block.  vmcnt retires in
// order, so "at most N younger operations outstanding" is exact as long as every iteration issues the same number of
// vector memory operations -- the fetches past the last block are therefore still issued, all lanes on one line.
__device__ __forceinline__ void gn_ld16(gn_h8& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void gn_ld16(f32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void gn_vmwait() { asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(N) : "memory"); }
// ties a prefetched register to the wait before it: no use of `r` is scheduled above this point
template <typename T>
__device__ __forceinline__ void gn_landed(T& r) { asm volatile("" : "+v"(r)); }
template <int R32, int D32, int KS>
__global__ __launch_bounds__(64 * (2 * D32 + R32 + 4)) void wn_gen_chain3_kernel(WnGenStepArgs a) {
  using S = GnShape<R32, D32, KS>;
  constexpr int R = S::R, D = S::D, JU = S::JU, KSR = S::KSR, KS0 = S::KS0, KS2 = S::KS2;
  constexpr int NS = 2;                               // register sets of the chain / conv1 waves (blocks in flight)
  constexpr int UB_BYTES = WN_GEN_CHAIN_MAX_BLOCKS * R * 4;   // conv1 biases of every block (filled once by the conv1 waves)
  constexpr int XOP_BYTES = KSR * 2048, ZOP_BYTES = KS2 * 2048;
  constexpr int TBL_BYTES = WN_GEN_CHAIN_MAX_BLOCKS * (int)sizeof(WnGenBlock);
  __shared__ __attribute__((aligned(16))) unsigned char smem[UB_BYTES + XOP_BYTES + 2 * ZOP_BYTES + TBL_BYTES];
  unsigned char* const ubuf = smem;
  unsigned char* const xop = smem + UB_BYTES;
  unsigned char* const zop = xop + XOP_BYTES;         // z operands of block b in half b & 1
  const WnGenBlock* const tbl = reinterpret_cast<const WnGenBlock*>(zop + 2 * ZOP_BYTES);   // valid after the first barrier
  const int lane = threadIdx.x & 63;
  // roles are per wave: the wave index goes through an SGPR so that the role branches are scalar branches (the roles'
  // code is then mutually exclusive for the compiler too, not a chain of exec-masked regions)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tl = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x;
  const int utt = tile * 32 + tl;
  const bool live = utt < a.B;
  const int ur = live ? utt : 0;
  const bool is_chain = wave < JU;
  const bool is_conv1 = wave >= JU && wave < JU + R32;
  const int cw = wave - JU;                           // conv1 tile of a conv1 wave
  const int sw = wave - JU - R32;                     // skip waves: column tiles 2 sw, 2 sw + 1
  const int nblocks = a.nblocks;
  if ((int)blockIdx.x >= a.ntiles) {
    // ---- helper workgroup: pulls the step's weight images into its XCD's L2 ----
    // L2 is cold at kernel start and a single CU keeps only so many misses in flight: the chain workgroup alone draws its
    // ~3.8 MB at ~65 GB/s, which is what bounds a step.  Workgroups go to the 8 XCDs round-robin by index, so the helpers
    // whose index is congruent to a chain workgroup's share its L2: they touch every 128 B line of every block's images in
    // block order (LDS-DMA dwords into a landing pad: no destination registers, nothing to wait for) and leave.  The chain
    // workgroup's own fetches then hit L2, or merge with a miss that is already on its way.
    const int hj = (int)blockIdx.x - a.ntiles, xcd = (int)blockIdx.x & 7, rank = hj >> 3;
    if (xcd >= a.ntiles) return;                      // no chain workgroup on this XCD
    int32_t* tb = reinterpret_cast<int32_t*>(zop + 2 * ZOP_BYTES);
    {
      const int32_t* src = reinterpret_cast<const int32_t*>(a.blocks);
      for (int i = threadIdx.x; i < nblocks * (int)(sizeof(WnGenBlock) / 4); i += blockDim.x) tb[i] = src[i];
    }
    __syncthreads();
    const int nw = (int)(blockDim.x >> 6);
    const int stride = WN_GEN_HELPERS_PER_XCD * nw * 64;          // lanes of all helpers of this XCD
    const int g0 = (rank * nw + wave) * 64 + lane;
    auto touch = [&](const void* base, int bytes) {
      const char* p = reinterpret_cast<const char*>(base);
      for (int l = g0; l * 128 < bytes; l += stride)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (int64_t)l * 128),
                                         (__attribute__((address_space(3))) void*)ubuf, 4, 0, 0);
    };
    for (int b = 0; b < nblocks; ++b) {
      const WnGenBlock& nb = tbl[b];
      touch(reinterpret_cast<const char*>(a.ws + nb.w16d_off) + (int64_t)KS0 * JU * 2048, KSR * JU * 2048);
      for (int t = xcd; t < a.ntiles; t += 8)
        touch(a.ws + a.u0_off + ((int64_t)b * a.ntiles + t) * (JU * 1024), JU * 4096);
      touch(a.ws + nb.w16r_off, KS2 * R32 * 2048);
      touch(a.params + nb.bias_r_off, R * 4);
      if (a.skip_tiles > 0)
        touch(reinterpret_cast<const char*>(a.ws + a.skip_w16_off) + (int64_t)b * KS2 * a.skip_tiles * 2048, KS2 * a.skip_tiles * 2048);
    }
    return;                                           // s_endpgm waits for what is still in flight
  }
#define GN_TS(role, b, k) do { if (a.ts && lane == 0 && (b) >= 8 && (b) < 12) a.ts[((role) * 4 + (b) - 8) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  // block table -> LDS; every role calls this AFTER issuing its first fetches (one cold round trip for both)
  auto copy_table = [&]() {
    const int32_t* src = reinterpret_cast<const int32_t*>(a.blocks);
    int32_t* dst = reinterpret_cast<int32_t*>(zop + 2 * ZOP_BYTES);
    for (int i = threadIdx.x; i < nblocks * (int)(sizeof(WnGenBlock) / 4); i += blockDim.x) dst[i] = src[i];
  };
  // tile of fp32 values (this lane's 16 accumulators = two k-steps) -> hi|lo B operands of k-steps 2j, 2j+1
  auto put_xop = [&](const f32x16& x, int j) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const f32x4 q0 = {x[8 * hf + 0], x[8 * hf + 1], x[8 * hf + 2], x[8 * hf + 3]};
      const f32x4 q1 = {x[8 * hf + 4], x[8 * hf + 5], x[8 * hf + 6], x[8 * hf + 7]};
      gn_h8 bh, bl;
      gn_split8(q0, q1, bh, bl);
      gn_h8* dst = reinterpret_cast<gn_h8*>(xop + (2 * j + hf) * 2048) + lane;
      dst[0] = bh;
      dst[64] = bl;
    }
  };

  if (is_chain) {
    // ================= chain waves: phase A and the gate, z k-step `wave` =================
    // Wave w owns a MIXED row tile of the gated conv: rows 0..15 = filter channels 16 w .. 16 w + 15, rows 16..31 = the
    // gate channels of the same 16 -- the two 16-row halves of image tiles w / 2 and D32 + w / 2.  A lane's 16 accumulators
    // are then 8 filter values and THEIR 8 gate values, the gate runs in registers, and the u tiles never go through LDS
    // (no barrier between phase A and the gate).  Only the lanes' fetch addresses differ from a plain tile: every output
    // element is the same dot product over the same operands in the same order.
    // per iteration: 2 KSR + 4 loads (fetch), then 2 stores (z row) -- see the wait in phase A
    gn_h8 wa[NS][KSR][2];
    f32x4 u0q[NS][4];                                 // the u0 images of the tile (dead columns read column 0's lines)
    const int ulane = live ? lane : h * 32;
    const int half16 = 16 * (wave & 1);               // which 16 rows of the image tiles
    const int wlane = (tl < 16 ? (wave >> 1) : D32 + (wave >> 1)) * 128 + half16 + (tl & 15) + 32 * h;   // tile * 128 + lane inside it
    auto fetch_a = [&](auto sc, int b, int64_t w16d_off) {
      constexpr int s = decltype(sc)::value;
      const bool real = b < nblocks;                  // past the end: every lane on one line of the parameters
      const gn_h8* wd = real ? reinterpret_cast<const gn_h8*>(a.ws + w16d_off) + (int64_t)KS0 * JU * 128 + wlane
                             : reinterpret_cast<const gn_h8*>(a.params);
      const int m = real ? 64 : 0;
#pragma unroll
      for (int kk = 0; kk < KSR; ++kk) {
        gn_ld16(wa[s][kk][0], wd + (kk * JU * 2 + 0) * m);
        gn_ld16(wa[s][kk][1], wd + (kk * JU * 2 + 1) * m);
      }
      // accumulator quads 0, 1 = quads 2 (w & 1), 2 (w & 1) + 1 of u0 tile w / 2; quads 2, 3 = the same of tile D32 + w / 2
      const f32x4* u0 = real ? reinterpret_cast<const f32x4*>(a.ws + a.u0_off) + (((int64_t)b * a.ntiles + tile) * (JU * 4) + (wave >> 1) * 4 + 2 * (wave & 1)) * 64 + ulane
                             : reinterpret_cast<const f32x4*>(a.params);
      gn_ld16(u0q[s][0], u0);
      gn_ld16(u0q[s][1], u0 + m);
      gn_ld16(u0q[s][2], u0 + D32 * 4 * m);
      gn_ld16(u0q[s][3], u0 + (D32 * 4 + 1) * m);
    };
    wn_static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      fetch_a(sc, s, a.blk0[s].w16d_off);
    });
    // every first fetch has landed before the loop is entered (block 0 needs its set anyway): whatever copies the
    // compiler places on the loop's entry edge then read settled registers
    gn_vmwait<0>();
    wn_static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
#pragma unroll
      for (int kk = 0; kk < KSR; ++kk) { gn_landed(wa[s][kk][0]); gn_landed(wa[s][kk][1]); }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) gn_landed(u0q[s][rq]);
    });
    copy_table();
    GN_BARRIER();
    for (int b0 = 0; b0 < nblocks; b0 += NS)
      wn_static_for<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int b = b0 + s;
        if (b < nblocks) {
          if (wave == 0) GN_TS(0, b, 0);
          gn_vmwait<(NS - 1) * (2 * KSR + 4)>();
          if (wave == 0) GN_TS(0, b, 1);
#pragma unroll
          for (int kk = 0; kk < KSR; ++kk) { gn_landed(wa[s][kk][0]); gn_landed(wa[s][kk][1]); }
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) gn_landed(u0q[s][rq]);
          f32x16 u;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            u[4 * rq + 0] = u0q[s][rq].x; u[4 * rq + 1] = u0q[s][rq].y; u[4 * rq + 2] = u0q[s][rq].z; u[4 * rq + 3] = u0q[s][rq].w;
          }
          gn_mac<KSR>(u, wa[s], reinterpret_cast<const gn_h8*>(xop) + lane);
          if (wave == 0) GN_TS(0, b, 2);
          // the gate: accumulators 0..7 are filter channels 16 w + {4h.., 8 + 4h..}, 8..15 their gate channels
          const f32x4 f0 = {u[0], u[1], u[2], u[3]}, f1 = {u[4], u[5], u[6], u[7]};
          const f32x4 g0 = {u[8], u[9], u[10], u[11]}, g1 = {u[12], u[13], u[14], u[15]};
          f32x4 z0, z1;
          z0.x = wn_tanh_fast(f0.x) * wn_sigmoid_fast(g0.x); z0.y = wn_tanh_fast(f0.y) * wn_sigmoid_fast(g0.y);
          z0.z = wn_tanh_fast(f0.z) * wn_sigmoid_fast(g0.z); z0.w = wn_tanh_fast(f0.w) * wn_sigmoid_fast(g0.w);
          z1.x = wn_tanh_fast(f1.x) * wn_sigmoid_fast(g1.x); z1.y = wn_tanh_fast(f1.y) * wn_sigmoid_fast(g1.y);
          z1.z = wn_tanh_fast(f1.z) * wn_sigmoid_fast(g1.z); z1.w = wn_tanh_fast(f1.w) * wn_sigmoid_fast(g1.w);
          gn_h8 bh, bl;
          gn_split8(z0, z1, bh, bl);
          gn_h8* zd = reinterpret_cast<gn_h8*>(zop + (b & 1) * ZOP_BYTES + wave * 2048) + lane;
          zd[0] = bh;
          zd[64] = bl;
          if (wave == 0) GN_TS(0, b, 4);
          GN_BARRIER();                               // (2) z operands visible
          if (wave == 0) GN_TS(0, b, 5);
          // the chain waves idle through phase C: the refill of set s (dead until block b + NS) is issued HERE -- a wave
          // that issues vector memory operations into a busy address pipeline stalls, which must not delay a barrier
          fetch_a(sc, b + NS, tbl[min(b + NS, nblocks - 1)].w16d_off);
          if (live) {                                 // gated activations of this block (fire and forget)
            float* zdst = a.ws + a.zrow_off + ((int64_t)b * a.B + utt) * D + 16 * wave + 4 * h;
            *reinterpret_cast<f32x4*>(zdst) = z0;
            *reinterpret_cast<f32x4*>(zdst + 8) = z1;
          }
          if (wave == 0) GN_TS(0, b, 6);
          GN_BARRIER();                               // (3) x operands visible
          if (wave == 0) GN_TS(0, b, 7);
        }
      });
    gn_vmwait<0>();                                   // the fetches past the last block: nothing may still be in flight
  } else if (is_conv1) {                              // when their registers are reused
    // (the conv1 waves carry the chain's critical path through phase C and share their SIMDs' matrix pipes with skip
    // waves, whose products have until the next block: issue priority to the conv1 waves -- 61.4 -> 59.3 us per step at
    // B = 8, same box; the chain waves' own priority makes no difference as long as it stays below)
    __builtin_amdgcn_s_setprio(3);
    // ================= conv1 waves: carry the block input x; phase C (o tile `cw`) =================
    // per iteration: 4 stores (ring) at the top, 2 KS2 loads (fetch) at the bottom -- see the wait in phase C
    gn_h8 wc[NS][KS2][2];
    float* const bias_l = reinterpret_cast<float*>(ubuf);   // [block][R]; wave cw fills and reads columns 32 cw .. 32 cw + 31
    auto fetch_c = [&](auto sc, bool real, int64_t w16r_off) {
      constexpr int s = decltype(sc)::value;
      const gn_h8* wr = real ? reinterpret_cast<const gn_h8*>(a.ws + w16r_off) + lane : reinterpret_cast<const gn_h8*>(a.params);
      const int m = real ? 64 : 0;
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        gn_ld16(wc[s][ks][0], wr + ((ks * R32 + cw) * 2 + 0) * m);
        gn_ld16(wc[s][ks][1], wr + ((ks * R32 + cw) * 2 + 1) * m);
      }
    };
    wn_static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      fetch_c(sc, s < nblocks, a.blk0[s].w16r_off);
    });
    // the conv1 biases of ALL blocks, once: 4 vector-memory operations a block and wave less than fetching them with the
    // fragments.  Wave cw keeps its own 32 columns (written and read by the same wave: LDS operations of a wave are ordered).
    // Requested here, with the first fragments, so that the kernel's start pays one cold round trip for both.
    constexpr int NBL = WN_GEN_CHAIN_MAX_BLOCKS * 32 / 64;
    float bl_[NBL];
    if (a.bias_r_stride != 0) {
#pragma unroll
      for (int q = 0; q < NBL; ++q) {
        const int i = lane + 64 * q, bb = min(i >> 5, nblocks - 1), c = 32 * cw + (i & 31);   // (odd block counts: clamped)
        if (64 * q < nblocks * 32) bl_[q] = a.params[a.bias_r_off0 + (int64_t)bb * a.bias_r_stride + c];   // wave-uniform
      }
    }
    gn_vmwait<0>();                                   // as in the chain role
    wn_static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) { gn_landed(wc[s][ks][0]); gn_landed(wc[s][ks][1]); }
    });
    copy_table();
    if (a.bias_r_stride != 0) {
#pragma unroll
      for (int q = 0; q < NBL; ++q) {
        const int i = lane + 64 * q;
        if (i < nblocks * 32) bias_l[(i >> 5) * R + 32 * cw + (i & 31)] = bl_[q];
      }
    } else {
      for (int i = lane; i < nblocks * 32; i += 64) {
        const int bb = i >> 5, c = 32 * cw + (i & 31);
        bias_l[bb * R + c] = a.params[a.blocks[bb].bias_r_off + c];
      }
    }
    // ---- input causal conv (C_in = 1): the k-ordered fma chain of the fp32 MFMA path, then + bias ----
    f32x16 x;
    {
      float xs[KS];
#pragma unroll
      for (int t = 0; t < KS; ++t) xs[t] = a.xin[(int)((unsigned)((int)a.tau - (KS - 1 - t)) % (unsigned)KS) * a.B + ur];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * cw + wn_drow(r, h);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) acc = fmaf(a.causal_w[t * R + c], xs[t], acc);
        x[r] = acc + a.causal_b[c];
      }
      put_xop(x, cw);
    }
    // range guard: running max-abs of the residual stream this wave carries (cast unscaled to fp16 hi | lo by put_xop)
    float xm = 0.f;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) xm = wn_absmax_acc(xm, x[4 * rq + 0], x[4 * rq + 1], x[4 * rq + 2], x[4 * rq + 3]);
    GN_BARRIER();
    for (int b0 = 0; b0 < nblocks; b0 += NS)
      wn_static_for<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int b = b0 + s;
        if (b < nblocks) {
          if (live) {                                 // this block's input at time tau -> its ring
            const WnGenBlock& cur = tbl[b];
            float* dst = a.ws + cur.ring_off + ((int)((unsigned)(int)a.tau % (unsigned)cur.nslots) * a.B + utt) * R + 32 * cw + 4 * h;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
              *reinterpret_cast<f32x4*>(dst + 8 * rq) = f32x4{x[4 * rq + 0], x[4 * rq + 1], x[4 * rq + 2], x[4 * rq + 3]};
          }
          if (cw == 0) GN_TS(1, b, 0);
          GN_BARRIER();                               // (2) z operands visible
          if (cw == 0) GN_TS(1, b, 2);
          gn_vmwait<(NS - 1) * 2 * KS2>();          // only LOADS count as younger: stores retire out of order with them
          if (cw == 0) GN_TS(1, b, 3);
#pragma unroll
          for (int ks = 0; ks < KS2; ++ks) { gn_landed(wc[s][ks][0]); gn_landed(wc[s][ks][1]); }
          f32x16 o;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + b * R + 32 * cw + 8 * rq + 4 * h);
            o[4 * rq + 0] = bv.x; o[4 * rq + 1] = bv.y; o[4 * rq + 2] = bv.z; o[4 * rq + 3] = bv.w;
          }
          gn_mac<KS2>(o, wc[s], reinterpret_cast<const gn_h8*>(zop + (b & 1) * ZOP_BYTES) + lane);
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = a.residual ? o[r] + x[r] : o[r];
          if (b + 1 < nblocks) put_xop(x, cw);
          if (cw == 0) GN_TS(1, b, 4);
          GN_BARRIER();                               // (3) x operands visible
          if (cw == 0) GN_TS(1, b, 5);
          // (range guard bookkeeping behind the hand-over: the conv1 waves idle through the next block's phases A and B)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) xm = wn_absmax_acc(xm, x[4 * rq + 0], x[4 * rq + 1], x[4 * rq + 2], x[4 * rq + 3]);
          {                                           // refill of set s, issued while the conv1 waves idle (phases A, B)
            fetch_c(sc, b + NS < nblocks, tbl[min(b + NS, nblocks - 1)].w16r_off);
          }
          if (cw == 0) GN_TS(1, b, 6);
        }
      });
    gn_vmwait<0>();                                   // the fetches past the last block
    // ---- the last block output feeds the head when use_skip is False ----
    if (a.hrow_off >= 0 && live) {
      float* dst = a.ws + a.hrow_off + (int64_t)utt * R + 32 * cw + 4 * h;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4*>(dst + 8 * rq) = f32x4{x[4 * rq + 0], x[4 * rq + 1], x[4 * rq + 2], x[4 * rq + 3]};
    }
    if (a.guard) wn_guard_publish_over(a.guard, live ? xm : 0.f);     // every step; an atomic only beyond the limit
  } else {
    // ================= skip waves: acc_t += W_{s,b}^T z for column tiles t0 = 2 sw and t1 = 2 sw + 1 =================
    // Two register sets per tile (set = block parity).  Tile t0 of block b runs between barriers (2) and (3) of block b,
    // tile t1 between (1) and (2) of block b + 1 -- z_b stays valid there because the z operands are double-buffered.
    const int t0 = 2 * sw, t1 = 2 * sw + 1;
    const bool has1 = t1 < a.skip_tiles;
    gn_h8 w0[KS2][2], w1[KS2][2];
    // vector memory operations of a skip wave: 2 KS2 loads per pre_skip, always issued (see the waits)
    auto pre_skip = [&](gn_h8 (&w)[KS2][2], int b, int t) {
      const bool real = b < nblocks;
      const gn_h8* wsi = real ? reinterpret_cast<const gn_h8*>(a.ws + a.skip_w16_off) + lane : reinterpret_cast<const gn_h8*>(a.params);
      const int64_t m = real ? 64 : 0;
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        const int64_t blk = ((int64_t)(b * KS2 + ks) * a.skip_tiles + t) * 2;
        gn_ld16(w[ks][0], wsi + (blk + 0) * m);
        gn_ld16(w[ks][1], wsi + (blk + 1) * m);
      }
    };
    // the other tile's fetch (if there is a second tile) is the only younger operation
    auto landed = [&](gn_h8 (&w)[KS2][2]) {
      if (has1) gn_vmwait<2 * KS2>(); else gn_vmwait<0>();
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) { gn_landed(w[ks][0]); gn_landed(w[ks][1]); }
    };
    auto mac = [&](f32x16& acc, const gn_h8 (&w)[KS2][2], int half) {
      gn_mac<KS2>(acc, w, reinterpret_cast<const gn_h8*>(zop + half * ZOP_BYTES) + lane);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // The two loops below contain no conditional fetch: a fetch under a branch makes the fetched registers a phi of two
    // definitions, and the copies that resolve it would read registers whose load is still in flight.
    auto settle = [&](gn_h8 (&w)[KS2][2]) {          // as in the chain role
      gn_vmwait<0>();
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) { gn_landed(w[ks][0]); gn_landed(w[ks][1]); }
    };
    if (has1 && a.skip_tiles <= 4) {
      // at most two skip waves: they sit on SIMDs 2 and 3, the conv1 waves on 0 and 1 -- both tiles run in phase C without
      // touching the conv1 waves' MFMA pipe, and nothing of the skip role stands between the chain waves' gate and
      // barrier (2).  Both refills in phase A.
      pre_skip(w0, 0, t0);
      pre_skip(w1, 0, t1);
      settle(w0);
      settle(w1);
      copy_table();
      GN_BARRIER();
      for (int b = 0; b < nblocks; ++b) {
        if (sw == 0) GN_TS(2, b, 3);
        GN_BARRIER();                                 // (2) z operands of block b visible
        if (sw == 0) GN_TS(2, b, 4);
        landed(w0);                                   // (younger: w1's fetch)
        mac(acc0, w0, b & 1);
        settle(w1);
        mac(acc1, w1, b & 1);
        if (sw == 0) GN_TS(2, b, 5);
        GN_BARRIER();                                 // (3)
        if (sw == 0) GN_TS(2, b, 6);
        pre_skip(w0, b + 1, t0);
        pre_skip(w1, b + 1, t1);
        if (sw == 0) GN_TS(2, b, 7);
      }
      gn_vmwait<0>();                                 // the fetches past the last block
    } else if (has1) {
      pre_skip(w0, 0, t0);
      pre_skip(w1, 0, t1);
      settle(w0);
      settle(w1);
      copy_table();
      GN_BARRIER();
      GN_BARRIER();                                   // (2) of block 0
      landed(w0);
      mac(acc0, w0, 0);
      GN_BARRIER();                                   // (3)
      pre_skip(w0, 1, t0);
      for (int b = 1; b < nblocks; ++b) {
        if (sw == 0) GN_TS(2, b, 1);
        landed(w1);                                   // tile t1 of block b - 1, then its refill (under the chain waves' phase A and gate)
        if (sw == 0) GN_TS(2, b, 2);
        mac(acc1, w1, (b - 1) & 1);
        pre_skip(w1, b, t1);
        if (sw == 0) GN_TS(2, b, 3);
        GN_BARRIER();                                 // (2) z operands of block b visible
        if (sw == 0) GN_TS(2, b, 4);
        landed(w0);
        mac(acc0, w0, b & 1);
        if (sw == 0) GN_TS(2, b, 5);
        GN_BARRIER();                                 // (3)
        if (sw == 0) GN_TS(2, b, 6);
        pre_skip(w0, b + 1, t0);                      // refill in phase A, not in phase C where the chain waves issue theirs
        if (sw == 0) GN_TS(2, b, 7);
      }
      gn_vmwait<0>();                                 // also the fetches past the last block
      landed(w1);                                     // tile t1 of the last block (its z half is not rewritten)
      mac(acc1, w1, (nblocks - 1) & 1);
    } else {
      pre_skip(w0, 0, t0);
      settle(w0);
      copy_table();
      GN_BARRIER();
      for (int b = 0; b < nblocks; ++b) {
        GN_BARRIER();                                 // (2) z operands of block b visible
        landed(w0);
        mac(acc0, w0, b & 1);
        GN_BARRIER();                                 // (3)
        pre_skip(w0, b + 1, t0);
      }
      gn_vmwait<0>();                                 // the fetch past the last block
    }
    // ---- folded skip sum + summed biases (the epilogue of the rows contraction it replaces) ----
    float sm = 0.f;
    if (live) {
      auto put = [&](const f32x16& acc, int t) {
        const float* bsp = a.ws + a.skip_bias_off + 32 * t + 4 * h;
        float* dst = a.ws + a.skiprow_off + (int64_t)utt * a.skip_ld + 32 * t + 4 * h;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bsp + 8 * rq);
          const f32x4 o = f32x4{wn_act(acc[4 * rq + 0] + bv.x, a.skip_act), wn_act(acc[4 * rq + 1] + bv.y, a.skip_act),
                                wn_act(acc[4 * rq + 2] + bv.z, a.skip_act), wn_act(acc[4 * rq + 3] + bv.w, a.skip_act)};
          *reinterpret_cast<f32x4*>(dst + 8 * rq) = o;
          sm = wn_absmax_acc(sm, o.x, o.y, o.z, o.w);
        }
      };
      put(acc0, t0);
      if (has1) put(acc1, t1);
    }
    if (a.guard) wn_guard_publish_over(a.guard, sm);  // the head casts this row to fp16 hi | lo
  }
}

// The head of a queued-generation step -- up to WN_GEN_HEAD_MAX 1x1 convs with bias and activation
// (src/model.py:105-119,237-238) -- for 32 utterances per workgroup in ONE launch instead of one launch
// per layer.  Wave w owns column tile w of the current layer (all layers here have at most 8 tiles); a
// layer's activated output is turned into fp16 hi | lo B operands (k-step = 16 channels) and handed to
// the next layer through LDS, exactly as the chain kernel hands x between blocks.  Per tile this is the
// thin rows GEMM's sequence -- k-steps in order, lo*hi, hi*lo, hi*hi, then acc + bias, activation -- so
// the logits are bit-identical to the per-layer launches (and to the sliding window).
__device__ __forceinline__ void gn_head_body(const WnGenHeadArgs& a, int tile, unsigned char* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tl = lane & 31, h = lane >> 5;
  const int utt = tile * 32 + tl;
  const bool live = utt < a.B;
  const int ur = live ? utt : 0;
  // ---- input rows -> operand buffer 0: wave w converts k-steps w, w + 8 ----
  {
    const float* row = a.ws + a.in_off + (int64_t)ur * a.in_ld + 4 * h;
    const int nks = a.K[0] / 16;
    for (int ks = wave; ks < nks; ks += 8) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(row + 16 * ks);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(row + 16 * ks + 8);
      gn_h8 bh, bl;
      gn_split8(q0, q1, bh, bl);
      gn_h8* dst = reinterpret_cast<gn_h8*>(smem + ks * 2048) + lane;
      dst[0] = bh;
      dst[64] = bl;
    }
  }
  __syncthreads();
  for (int li = 0; li < a.nlayers; ++li) {
    const int nks = a.K[li] / 16, nt = a.N[li] / 32;
    const unsigned char* ib = smem + (li & 1) * (16 * 2048);
    unsigned char* ob = smem + ((li + 1) & 1) * (16 * 2048);
    const bool to_f32 = li + 1 == a.nlayers && a.f32_K > 0;      // its output feeds the exact-fp32 last layer
    const bool lastl = li + 1 == a.nlayers && !to_f32;
    if (wave < nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const gn_h8* wimg = reinterpret_cast<const gn_h8*>(a.ws + a.w16_off[li]) + lane;
      const gn_h8* xl = reinterpret_cast<const gn_h8*>(ib) + lane;
      // weight fragments four k-steps ahead (register ring, static slots)
      gn_h8 wf[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t blk = ((int64_t)min(i, nks - 1) * nt + wave) * 2;
        wf[i][0] = wimg[(blk + 0) * 64];
        wf[i][1] = wimg[(blk + 1) * 64];
      }
      for (int ks0 = 0; ks0 < nks; ks0 += 4) {
        wn_static_for<4>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (ks0 + i < nks) {                          // wave-uniform
            const gn_h8 bh = xl[((ks0 + i) * 2 + 0) * 64];
            const gn_h8 bl = xl[((ks0 + i) * 2 + 1) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][1], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][0], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][0], bh, acc, 0, 0, 0);
            const int64_t blk = ((int64_t)min(ks0 + i + 4, nks - 1) * nt + wave) * 2;
            wf[i][0] = wimg[(blk + 0) * 64];
            wf[i][1] = wimg[(blk + 1) * 64];
          }
        });
      }
      // epilogue of the rows GEMM: acc (* 1) + bias, activation
      const float* bias = a.params + a.bias_off[li] + 32 * wave + 4 * h;
      float v[16];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 8 * rq);
        v[4 * rq + 0] = wn_act(acc[4 * rq + 0] + bv.x, a.act[li]);
        v[4 * rq + 1] = wn_act(acc[4 * rq + 1] + bv.y, a.act[li]);
        v[4 * rq + 2] = wn_act(acc[4 * rq + 2] + bv.z, a.act[li]);
        v[4 * rq + 3] = wn_act(acc[4 * rq + 3] + bv.w, a.act[li]);
      }
      if (lastl) {
        if (live) {
          float* dst = a.ws + a.out_off + (int64_t)utt * a.N[li] + 32 * wave + 4 * h;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4*>(dst + 8 * rq) = f32x4{v[4 * rq + 0], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
        }
      } else {
        // this tile = k-steps 2 wave, 2 wave + 1 of the next layer (cast to fp16 hi | lo unscaled: range guard)
        if (a.guard) {
          float vm = 0.f;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) vm = wn_absmax_acc(vm, v[4 * rq + 0], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]);
          wn_guard_publish_over(a.guard, live ? vm : 0.f);
        }
        if (to_f32) {
          // ... or k-quads 4 wave .. 4 wave + 3 of the fp32 layer: accumulator quad rq of a lane IS that lane's B operand
          // of quad 4 wave + rq (channels 8 q + 4 h .. + 3 of its row), kept in fp32
          f32x4* xf = reinterpret_cast<f32x4*>(ob);
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            xf[(4 * wave + rq) * 64 + lane] = f32x4{v[4 * rq + 0], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
        } else
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 q0 = {v[8 * hf + 0], v[8 * hf + 1], v[8 * hf + 2], v[8 * hf + 3]};
          const f32x4 q1 = {v[8 * hf + 4], v[8 * hf + 5], v[8 * hf + 6], v[8 * hf + 7]};
          gn_h8 bh, bl;
          gn_split8(q0, q1, bh, bl);
          gn_h8* dst = reinterpret_cast<gn_h8*>(ob + (2 * wave + hf) * 2048) + lane;
          dst[0] = bh;
          dst[64] = bl;
        }
      }
    }
    __syncthreads();
  }
  // ---- exact-fp32 last layer (at most 32 columns: one row tile, one wave): wn_gemm_rows_kernel<1>'s arithmetic -- k-quads
  // ascending, four v_mfma_f32_32x32x2_f32 a quad into one accumulator from zero, then + bias -- so the outputs are the
  // per-layer launch's and the sliding window's.  Fragments eight quads ahead (a quad is 4 dependent products = 256 clocks).
  if (a.f32_K > 0) {
    float* lg = reinterpret_cast<float*>(smem + ((a.nlayers & 1) ^ 1) * (16 * 2048));   // [32 rows][32]: the buffer the last layer read
    if (wave == 0) {
      const f32x4* xf = reinterpret_cast<const f32x4*>(smem + (a.nlayers & 1) * (16 * 2048)) + lane;   // written by the last layer
      const f32x4* fr = reinterpret_cast<const f32x4*>(a.ws + a.f32_w_off) + lane;    // [quad][64 lanes] (one row tile)
      const int nq = a.f32_K / 8;
      constexpr int PF = 8;
      f32x4 ar[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) ar[i] = fr[(int64_t)min(i, nq - 1) * 64];
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int q0 = 0; q0 < nq; q0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          f32x4 xv = xf[(q0 + i) * 64];
          if (!live) xv = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 av = ar[i];
          acc = wn_mfma(av.x, xv.x, acc);
          acc = wn_mfma(av.y, xv.y, acc);
          acc = wn_mfma(av.z, xv.z, acc);
          acc = wn_mfma(av.w, xv.w, acc);
          ar[i] = fr[(int64_t)min(q0 + i + PF, nq - 1) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const float* bias = a.params + a.f32_bias_off;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 8 * rq + 4 * h + e;
          if (n < a.f32_N) {
            const float w = acc[4 * rq + e] + bias[n];
            lg[tl * 32 + n] = w;
            if (live) a.ws[a.out_off + (int64_t)utt * a.f32_N + n] = w;
          }
        }
    }
    __syncthreads();
    // mixture sampling tail and emit, one row per thread: the rows of wn_sample_det_mix_kernel / wn_sample_rand_mix_kernel
    if (a.tail >= 3 && threadIdx.x < 32) {
      const int row = tile * 32 + (int)threadIdx.x;
      if (row < a.B) {
        const float* p = lg + threadIdx.x * 32;
        const float v = a.tail == 3 ? wn_mix_det_row(p, a.mix_M) : wn_mix_rand_row(p, a.mix_M, a.mix_kind, row, a.seed, a.offset);
        if (a.samp) a.samp[row] = v;
        a.em.out[(int64_t)row * a.em.length + a.em.step] = v;
        if (a.em.xin_slot) a.em.xin_slot[row] = v;
      }
    }
    return;
  }
  // ---- categorical sampling tail (softmax -> arg max, or softmax -> inverse-CDF draw) and emit, one wave per row: the
  // rows of wn_gen_tail_cat_det_kernel / wn_sample_rand_cat_logits_kernel, so the samples are theirs.  The logits were
  // stored by other waves of this workgroup: the barrier above has drained those stores, and nothing has read these
  // lines into this CU's L1 before ----
  if (a.tail == 1 || a.tail == 2) {
    const int C = a.N[a.nlayers - 1];
    float* q = reinterpret_cast<float*>(smem) + wave * 256;     // the operand buffers are free now (C <= 256)
    for (int i = wave; i < 32; i += 8) {
      const int row = tile * 32 + i;
      if (row >= a.B) break;                          // wave-uniform
      const float* l = a.ws + a.out_off + (int64_t)row * C;
      const float v = a.tail == 1 ? wn_cat_det_row(l, C, lane, a.inv_lv)
                                  : wn_cat_rand_row(l, C, lane, q, row, a.seed, a.offset, a.inv_lv);
      if (lane == 0) {
        if (a.samp) a.samp[row] = v;
        a.em.out[(int64_t)row * a.em.length + a.em.step] = v;
        if (a.em.xin_slot) a.em.xin_slot[row] = v;
      }
    }
  }
}
__global__ __launch_bounds__(512) void wn_gen_head_kernel(WnGenHeadArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16 * 2048];       // two operand buffers of 16 k-steps
  gn_head_body(a, (int)blockIdx.x, smem);
}
// The head of step tau and, in further workgroups of the same launch, the pre kernel's work for step tau + 1: that part
// reads rings only, all written once the chain kernel of step tau has finished, so it does not have to wait for the
// sample -- one launch (and its gap) less per step.
template <int R32, int D32, int KS>
__global__ __launch_bounds__(512) void wn_gen_head_pre_kernel(WnGenHeadArgs ha, WnGenStepArgs ga, int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16 * 2048];
  if ((int)blockIdx.x < ntiles) {
    gn_head_body(ha, (int)blockIdx.x, smem);
  } else {
    if (threadIdx.x >= 64) return;                    // one wave per (tile, block), as in the stand-alone launch
    const int j = (int)blockIdx.x - ntiles;
    gn_pre_body<R32, D32, KS>(ga, j % ntiles, j / ntiles, ntiles);
  }
}

static int gn_head_check(const WnGenHeadArgs& a) {
  if (a.nlayers < 1 || a.nlayers > WN_GEN_HEAD_MAX) { wn_set_error("gen_head: bad layer count"); return WN_E_UNSUPPORTED; }
  for (int i = 0; i < a.nlayers; ++i)
    if (a.K[i] % 16 != 0 || a.K[i] > 256 || a.N[i] % 32 != 0 || a.N[i] > 256 || (i > 0 && a.K[i] != a.N[i - 1])) {
      wn_set_error("gen_head: unsupported layer shape");
      return WN_E_UNSUPPORTED;
    }
  return WN_OK;
}
int wn_launch_gen_head_pre(const WnGenHeadArgs& a, const WnGenStepArgs& g, int R, int KS, hipStream_t s) {
  const int rc = gn_head_check(a);
  if (rc) return rc;
  const int nt = (a.B + 31) / 32;
  const dim3 grid((unsigned)(nt + nt * g.nblocks));
  if (R == 32 && KS == 2) hipLaunchKernelGGL((wn_gen_head_pre_kernel<1, 1, 2>), grid, dim3(512), 0, s, a, g, nt);
  else if (R == 32 && KS == 3) hipLaunchKernelGGL((wn_gen_head_pre_kernel<1, 1, 3>), grid, dim3(512), 0, s, a, g, nt);
  else if (R == 64 && KS == 2) hipLaunchKernelGGL((wn_gen_head_pre_kernel<2, 2, 2>), grid, dim3(512), 0, s, a, g, nt);
  else { wn_set_error("gen_head_pre: unsupported shape"); return WN_E_UNSUPPORTED; }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int wn_launch_gen_head(const WnGenHeadArgs& a, hipStream_t s) {
  if (a.nlayers < 1 || a.nlayers > WN_GEN_HEAD_MAX) { wn_set_error("gen_head: bad layer count"); return WN_E_UNSUPPORTED; }
  for (int i = 0; i < a.nlayers; ++i)
    if (a.K[i] % 16 != 0 || a.K[i] > 256 || a.N[i] % 32 != 0 || a.N[i] > 256 || (i > 0 && a.K[i] != a.N[i - 1])) {
      wn_set_error("gen_head: unsupported layer shape");
      return WN_E_UNSUPPORTED;
    }
  if (a.f32_K > 0 && (a.f32_K != a.N[a.nlayers - 1] || a.f32_K % 64 != 0 || a.f32_N < 1 || a.f32_N > 32 ||
                      (a.tail >= 3 && 3 * a.mix_M != a.f32_N))) {
    wn_set_error("gen_head: unsupported fp32 last layer");
    return WN_E_UNSUPPORTED;
  }
  if (a.f32_K == 0 && a.tail >= 3) { wn_set_error("gen_head: the mixture tail follows the fp32 layer"); return WN_E_UNSUPPORTED; }
  hipLaunchKernelGGL(wn_gen_head_kernel, dim3((unsigned)((a.B + 31) / 32)), dim3(512), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// the chain kernel keeps a copy of the block table and every conv1 bias in LDS: at most this many blocks per launch
int wn_gen_chain_max_blocks() { return WN_GEN_CHAIN_MAX_BLOCKS; }

int wn_gen_blocks_supported(int R, int D, int KS) {
  if (R == 32 && D == 32) return KS == 2 || KS == 3;
  if (R == 64 && D == 64) return KS == 2;
  return 0;
}

// the chain kernel can carry the folded skip contraction when its output is at most 8 column tiles
int wn_gen_skip_fusable(int S) { return S % 32 == 0 && S / 32 >= 1 && S / 32 <= 8; }

int64_t wn_gen_u0_floats(int B, int nblocks, int D) {
  return (int64_t)nblocks * ((B + 31) / 32) * (2 * D / 32) * 1024;
}

static unsigned long long* g_gen_ts = nullptr;
static unsigned long long* wn_gen_ts_buffer() {
  if (!g_gen_ts) { (void)hipMalloc((void**)&g_gen_ts, 3 * 4 * 8 * 8); (void)hipMemset(g_gen_ts, 0, 3 * 4 * 8 * 8); }
  return g_gen_ts;
}
extern "C" int wn_debug_gen_ts(unsigned long long* out) {
  if (!g_gen_ts) return -1;
  return (int)hipMemcpy(out, g_gen_ts, 3 * 4 * 8 * 8, hipMemcpyDeviceToHost);
}
template <int R32, int D32, int KS>
static void gn_launch(const WnGenStepArgs& a, int what, hipStream_t s) {   // what: 1 pre, 2 chain, 3 both
  const unsigned gx = (unsigned)((a.B + 31) / 32);
  if (what & 1) hipLaunchKernelGGL((wn_gen_pre_kernel<R32, D32, KS>), dim3(gx, (unsigned)a.nblocks), dim3(64), 0, s, a);
  if (!(what & 2)) return;
  WnGenStepArgs a2 = a;
  a2.ts = wn_debug_get(24) ? wn_gen_ts_buffer() : nullptr;      // knob 24: s_memtime phase stamps (profiling hook)
  a2.ntiles = (int)gx;
  const unsigned helpers = 8u * WN_GEN_HELPERS_PER_XCD;         // workgroups that pull the next images into every XCD's L2
  hipLaunchKernelGGL((wn_gen_chain3_kernel<R32, D32, KS>), dim3(gx + helpers), dim3(64 * (2 * D32 + R32 + (a.skip_tiles + 1) / 2)), 0, s, a2);
}

// what: 1 = the pre kernel (older taps of every block for time a.tau: reads rings only, so it may run as soon as the
// chain kernel of time a.tau - 1 has finished), 2 = the chain kernel, 3 = both in order
int wn_launch_gen_blocks(const WnGenStepArgs& a, int R, int KS, int what, hipStream_t s) {
  if (a.nblocks > WN_GEN_CHAIN_MAX_BLOCKS) { wn_set_error("gen_blocks: more than %d blocks", WN_GEN_CHAIN_MAX_BLOCKS); return WN_E_UNSUPPORTED; }
  if (R == 32 && KS == 2) gn_launch<1, 1, 2>(a, what, s);
  else if (R == 32 && KS == 3) gn_launch<1, 1, 3>(a, what, s);
  else if (R == 64 && KS == 2) gn_launch<2, 2, 2>(a, what, s);
  else { wn_set_error("gen_blocks: unsupported shape"); return WN_E_UNSUPPORTED; }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
