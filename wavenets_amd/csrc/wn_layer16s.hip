// Fused WaveNet residual block forward for blocks whose weights do not fit LDS (R = D = 128), split-precision MFMA,
// STREAMED weights (gfx950).   Reference: WaveNetLayer.call, src/layers.py:178-224 (depth-1 dilated stack).
//
// Same chain and register orientation as wn_layer16.hip -- time on lanes, u = b_d + sum_tap W_tap^T x[t - shift] ->
// z = tanh * sigmoid -> o = b_r + W_r^T z -> x_out = o + x, every product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
// v_mfma_f32_32x32x16_f16 -- but the fp16 hi|lo images of one block are 256 KiB (gated conv) + 64 KiB (1x1), twice the LDS.
// So the workgroup (8 waves = 256 rows per pass) STREAMS them: the images are one sequence of 20 chunks of 16 KiB (16
// k-steps of the conv with all 8 row tiles, then 4 x two k-steps of the 1x1) that cycles through a 4-deep LDS ring
// filled by LDS-DMA, two chunks in flight, one raw barrier per chunk; the stream never stops at a tile boundary (the
// weights do not depend on the tile).  The activations of a k-step come by LDS-DMA too (three k-steps in flight, into the
// wave's otherwise idle output stage), so the only waits in the loop are counted s_waitcnt vmcnt(N).  u (32 x 256) stays
// in 128 accumulator registers per lane, the gate runs in registers and z feeds the 1x1 as the B operand as it stands: u
// never touches HBM and z is written once (for the folded skip contraction and backward), never read back here.
//
// Measured skeleton (tools/stream_probe.hip: the same DMA / barrier / store structure without the arithmetic): the L2-resident
// weight stream costs nothing beside the HBM streams (48 B/clk/CU alone, +3 % beside them) and the structure runs at the
// HBM rate of its activation traffic (5.2 TB/s).
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const f32x4& q0, const f32x4& q1, h8& hi, h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)v[e];
    hi[e] = h;
    lo[e] = (_Float16)(v[e] - (float)h);
  }
}

// LDS-DMA of 16 bytes per lane: global address = scalar base + 32-bit lane offset, LDS address = M0 + lane * 16.
// Written as inline assembly on purpose: through the builtin hipcc forms every address as a 64-bit VGPR pair, hoists the
// pairs of all 22 requests of a tile out of the tile loop, spills them, and reloads each one with s_waitcnt vmcnt(0) in front
// of its request -- which drains the pipeline this kernel is built around.  (The compiler does not count these requests
// in its own vmcnt bookkeeping: its waits for its own loads only become more conservative.)
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");   // (m0 is reserved: the compiler never keeps a value in it)
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

template <int KS>
struct G {
  static constexpr int R = 128, D = 128, JU = 8, R32 = 4, D32 = 4;
  static constexpr int NC1 = KS * R / 16;              // conv chunks (one k-step of 8 row tiles = 16 KiB each)
  static constexpr int NC2 = (D / 16) * R32 / 8;       // 1x1 chunks (two k-steps of 4 row tiles each)
  static constexpr int NCH = NC1 + NC2;
  static constexpr int CHUNK = 16384, NBUF = 4, XB = 4, XBUF = 2048;
  static constexpr int PITCH = 68, STAGE = 32 * PITCH * 4;          // 8704 >= XB * XBUF
  static constexpr int WAVES = 8;
  static constexpr int BIAS = (2 * D + R) * 4;
  static constexpr int LDS = NBUF * CHUNK + WAVES * STAGE + BIAS;   // 136704
  static constexpr int PT = CHUNK / 16 / 512;          // weight-DMA instructions per thread and chunk
  static constexpr int PX = 2;                         // activation-DMA instructions per lane and k-step
};

// 32 x 64 half tile (two D-layout accumulator tiles) -> wave-private LDS stage -> 256-byte row segments in HBM.
// dst = wave-uniform base of the tile's first row (+ column offset), voff = this lane's byte offset inside a group of four
// rows: the stores are scalar base + 32-bit lane offset (64-bit per-row addresses in VGPRs are what spills in the epilogue);
// the read-back runs in two groups of four rows so that at most 16 data registers are in flight.
template <int PITCH>
__device__ __forceinline__ void store_half(const f32x16& v0, const f32x16& v1, float* stage, float* dst, unsigned voff,
                                           unsigned ld_bytes, int rows_valid, int lane) {
  const int tl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x16& v = jj ? v1 : v0;
      f32x4 o;
      o.x = v[4 * rq + 0]; o.y = v[4 * rq + 1]; o.z = v[4 * rq + 2]; o.w = v[4 * rq + 3];
      *reinterpret_cast<f32x4*>(stage + tl * PITCH + 32 * jj + 8 * rq + 4 * h) = o;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float* rd = stage + (lane >> 4) * PITCH + (lane & 15) * 4;
  char* base = reinterpret_cast<char*>(dst);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int i = 4 * g; i < 4 * g + 4; ++i) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(rd + i * 4 * PITCH);
      if (i * 4 + (lane >> 4) < rows_valid)
        *reinterpret_cast<f32x4*>(base + (uint64_t)((unsigned)(i * 4) * ld_bytes) + voff) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("" ::: "memory");
}

// RESMODE 0: no residual; 1: x_out = o + (a.res ? a.res : a.x).  SAVE: the sigmoid is written for backward (training).
template <int KS, int RESMODE, bool SAVE>
__global__ __launch_bounds__(512, 2) void wn_layer_fwd_s128_kernel(WnLayerFwdArgs a) {
  using C = G<KS>;
  constexpr int R = C::R, D = C::D, NC1 = C::NC1, NCH = C::NCH, PITCH = C::PITCH, PT = C::PT, PX = C::PX;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  const int tid = threadIdx.x;
  // the wave index through an SGPR: every LDS-DMA destination (M0) and every ring address is then scalar arithmetic
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  float* stage = reinterpret_cast<float*>(smem + C::NBUF * C::CHUNK + wave * C::STAGE);
  unsigned char* const xbuf = reinterpret_cast<unsigned char*>(stage);
  const unsigned smem_addr = lds_addr_of(smem), xbuf_addr = lds_addr_of(xbuf);
  float* sbias = reinterpret_cast<float*>(smem + C::NBUF * C::CHUNK + C::WAVES * C::STAGE);
  for (int i = tid; i < 2 * D; i += 512) sbias[i] = a.bias_d[i];
  for (int i = tid; i < R; i += 512) sbias[2 * D + i] = a.bias_r[i];
  const float* lbias_d = sbias;
  const float* lbias_r = sbias + 2 * D;

  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  const int64_t per_pass = (int64_t)gridDim.x * C::WAVES;
  const int64_t passes = (ntiles + per_pass - 1) / per_pass;

  // weight chunk of stream position g (chunk g % NCH of the images) -> ring buffer g % NBUF; every wave instruction
  // moves one contiguous KiB (wave-uniform LDS base + lane * 16)
  // (source = scalar base + ONE 32-bit per-thread byte offset: 64-bit per-chunk addresses in VGPRs get hoisted out of the
  // tile loop and spilled)
  const unsigned woff = (unsigned)tid * 16u;
  auto wdma = [&](int cc, int slot) {                  // cc = chunk inside the tile's sequence, compile-time at every call
    const char* base = cc < NC1 ? reinterpret_cast<const char*>(a.frag_d) + (int64_t)cc * C::CHUNK
                                : reinterpret_cast<const char*>(a.frag_r) + (int64_t)(cc - NC1) * C::CHUNK;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      dma16(base + 8192 * i, woff, smem_addr + slot * C::CHUNK + (512 * i + wave * 64) * 16);
    }
  };

  float wmax = 0.f;
  __syncthreads();                                     // bias table
  // the first two chunks of the stream
  wdma(0, 0);
  wdma(1, 1);

  for (int64_t pass = 0; pass < passes; ++pass) {
    const int64_t tile = (pass * gridDim.x + blockIdx.x) * C::WAVES + wave;
    const bool live = tile < ntiles;                   // dead waves still take part in the barriers and the weight stream
    const int b = live ? (int)(tile / tiles_per_b) : 0;
    const int t0 = live ? (int)(tile % tiles_per_b) * 32 : 0;
    const int t = t0 + tl;
    const bool tin = live && t < a.T;
    const int rows_valid = live ? min(32, a.T - t0) : 0;
    const int64_t row0 = (int64_t)b * a.T + t0;

    // per-tap source row of this lane (clamped; masked rows are zeroed at use)
    // (32-bit byte offsets from a scalar base; the launcher checks that the tensors stay below 4 GiB)
    unsigned xoff[KS];
    bool xok[KS];
#pragma unroll
    for (int tap = 0; tap < KS; ++tap) {
      const int ts = a.xt[tap] ? t : t - (KS - 1 - tap) * a.dilation;
      xok[tap] = tin && ts >= 0;
      xoff[tap] = (unsigned)(((int64_t)b * a.T + (xok[tap] ? ts : 0)) * R + 4 * h) * 4u;
    }
    auto xdma = [&](int c) {                           // activations of conv k-step c -> activation buffer c % XB
      const int tap = c / (R / 16), kk = c % (R / 16);
      const char* base = reinterpret_cast<const char*>(a.xt[tap] ? a.xt[tap] : a.x) + 64 * kk;
      const unsigned dst = xbuf_addr + (c & (C::XB - 1)) * C::XBUF;  // wave-uniform
      dma16(base, xoff[tap], dst);
      dma16(base + 32, xoff[tap], dst + 1024);
    };

    // ---- accumulators start at the bias (+ per-utterance conditioning bias) ----
    f32x16 u[C::JU];
#pragma unroll
    for (int j = 0; j < C::JU; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias_d + 32 * j + 8 * rq + 4 * h);
        u[j][4 * rq + 0] = bv.x; u[j][4 * rq + 1] = bv.y; u[j][4 * rq + 2] = bv.z; u[j][4 * rq + 3] = bv.w;
      }
    if (a.cb) {   // wave-uniform
      const float* cbp = a.cb + (int64_t)b * 2 * D + 4 * h;
#pragma unroll
      for (int j = 0; j < C::JU; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 cv = *(const __attribute__((address_space(1))) f32x4*)(cbp + 32 * j + 8 * rq);
          u[j][4 * rq + 0] += cv.x; u[j][4 * rq + 1] += cv.y; u[j][4 * rq + 2] += cv.z; u[j][4 * rq + 3] += cv.w;
        }
    }
    // Issue order per tile:  x0 x1 x2 | w(c+2) x(c+3) | ...   with w(0), w(1) of THIS tile issued during the previous
    // tile's last two chunks (or before the loop).  vmcnt retires loads in order, so "at most N outstanding" with
    // N = the LOADS issued after the ones chunk c needs means they have landed (stores in between only make the wait
    // conservative: they never count as younger loads).
    xdma(0);
    xdma(1);
    xdma(2);
    static_assert(NCH % C::NBUF == 0, "a tile's chunk sequence must start at ring slot 0");
    constexpr int ring0 = 0;                             // NCH % NBUF == 0: every tile's chunk c lives in slot c % NBUF
    // =================== dilated causal conv: NC1 chunks ===================
    wn_static_for<NC1>([&](auto cc_) {
      constexpr int c = decltype(cc_)::value;
      wdma((c + 2) % NCH, (ring0 + c + 2) & (C::NBUF - 1));      // (past the tile's end: the next tile's first chunks)
      if constexpr (c + 3 < NC1) xdma(c + 3);
      // loads issued after w(c) / x(c):   [x(c+1) x(c+2)] w(c+1) w(c+2) [x(c+3)]  as far as they exist
      constexpr int nx = (c + 1 < NC1) + (c + 2 < NC1) + (c + 3 < NC1);
      constexpr int nw = c == 0 ? 1 : 2;               // at c == 0 the tile's w(0), w(1) are older than x(0)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nw * PT + nx * PX) : "memory");
      asm volatile("s_barrier" ::: "memory");
      const h8* wl = reinterpret_cast<const h8*>(smem + ((ring0 + c) & (C::NBUF - 1)) * C::CHUNK) + lane;
      const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + (c & (C::XB - 1)) * C::XBUF) + lane;
      constexpr int tap = c / (R / 16);
      f32x4 q0 = xl[0], q1 = xl[64];
      h8 fr[2][2];
      fr[0][0] = wl[0];
      fr[0][1] = wl[64];
      if (!xok[tap]) { q0 = f32x4{0.f, 0.f, 0.f, 0.f}; q1 = q0; }
      h8 bh, bl;
      split8(q0, q1, bh, bl);
      wn_static_for<C::JU>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j + 1 < C::JU) {
          fr[(j + 1) & 1][0] = wl[((j + 1) * 2 + 0) * 64];
          fr[(j + 1) & 1][1] = wl[((j + 1) * 2 + 1) * 64];
        }
        u[j] = mfma16(fr[j & 1][1], bh, u[j]);
        u[j] = mfma16(fr[j & 1][0], bl, u[j]);
        u[j] = mfma16(fr[j & 1][0], bh, u[j]);
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this chunk's buffers are free again
    });

    // =================== gate (in place): u[j] -> z, u[j + 4] -> sigmoid ===================
#pragma unroll
    for (int j = 0; j < C::D32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sg = wn_sigmoid_fast(u[j + C::D32][r]);
        u[j + C::D32][r] = sg;
        u[j][r] = wn_tanh_fast(u[j][r]) * sg;
      }
    // lane offset inside a group of four rows (row stride 512 B for the R / D wide tensors, ldz * 4 for z)
    const unsigned voff128 = (unsigned)(lane >> 4) * 512u + (unsigned)(lane & 15) * 16u;
    if (rows_valid > 0) {
      if constexpr (SAVE) {
        store_half<PITCH>(u[4], u[5], stage, a.ag_out + row0 * D, voff128, 512u, rows_valid, lane);
        store_half<PITCH>(u[6], u[7], stage, a.ag_out + row0 * D + 64, voff128, 512u, rows_valid, lane);
      }
      if (a.z_out) {
        const unsigned ldzb = (unsigned)a.ldz * 4u;
        const unsigned voffz = (unsigned)(lane >> 4) * ldzb + (unsigned)(lane & 15) * 16u;
        store_half<PITCH>(u[0], u[1], stage, a.z_out + row0 * a.ldz, voffz, ldzb, rows_valid, lane);
        store_half<PITCH>(u[2], u[3], stage, a.z_out + row0 * a.ldz + 64, voffz, ldzb, rows_valid, lane);
      }
    }

    // The residual (x itself, or a separate tensor: dropout feeds the conv a dropped copy, queued generation ring rows) is
    // re-read here in D layout, ahead of the 1x1 whose products hide its latency: keeping the newest tap's 64 registers
    // alive through the conv does not fit beside the 128 accumulators (the compiler spills all of them).
    f32x4 xr[RESMODE == 1 ? C::R32 : 1][4];
    if constexpr (RESMODE == 1) {
      const char* rbase = reinterpret_cast<const char*>(a.res ? a.res : a.x) + row0 * (R * 4);
      const unsigned roff = (unsigned)(tin ? tl : 0) * (R * 4u) + 16u * h;
#pragma unroll
      for (int j = 0; j < C::R32; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          xr[j][rq] = *(const __attribute__((address_space(1))) f32x4*)(rbase + (128 * j + 32 * rq) + roff);
    }
    // =================== 1x1 residual conv: NC2 chunks of two k-steps; B operand = the z tiles as they stand ===================
    f32x16 o[C::R32];
#pragma unroll
    for (int j = 0; j < C::R32; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias_r + 32 * j + 8 * rq + 4 * h);
        o[j][4 * rq + 0] = bv.x; o[j][4 * rq + 1] = bv.y; o[j][4 * rq + 2] = bv.z; o[j][4 * rq + 3] = bv.w;
      }
    wn_static_for<C::NC2>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      constexpr int c = NC1 + cc;
      // (the last pass's two look-ahead chunks are re-reads of chunks 0, 1 nobody uses: the counted waits stay the same)
      wdma((c + 2) % NCH, (ring0 + c + 2) & (C::NBUF - 1));
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
      asm volatile("s_barrier" ::: "memory");
      const h8* wl = reinterpret_cast<const h8*>(smem + ((ring0 + c) & (C::NBUF - 1)) * C::CHUNK) + lane;
      h8 fr[2][2];
      fr[0][0] = wl[0];
      fr[0][1] = wl[64];
      wn_static_for<2>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int ks = 2 * cc + k;
        constexpr int jz = ks / 2, r0 = 8 * (ks % 2);
        f32x4 q0, q1;
        q0.x = u[jz][r0 + 0]; q0.y = u[jz][r0 + 1]; q0.z = u[jz][r0 + 2]; q0.w = u[jz][r0 + 3];
        q1.x = u[jz][r0 + 4]; q1.y = u[jz][r0 + 5]; q1.z = u[jz][r0 + 6]; q1.w = u[jz][r0 + 7];
        h8 bh, bl;
        split8(q0, q1, bh, bl);
        wn_static_for<C::R32>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int blk = k * C::R32 + j;
          if constexpr (blk + 1 < 2 * C::R32) {
            fr[(blk + 1) & 1][0] = wl[((blk + 1) * 2 + 0) * 64];
            fr[(blk + 1) & 1][1] = wl[((blk + 1) * 2 + 1) * 64];
          }
          o[j] = mfma16(fr[blk & 1][1], bh, o[j]);
          o[j] = mfma16(fr[blk & 1][0], bl, o[j]);
          o[j] = mfma16(fr[blk & 1][0], bh, o[j]);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });

    // =================== residual, range guard, x_out ===================
    if (rows_valid > 0) {
      if (a.o_out) {
        store_half<PITCH>(o[0], o[1], stage, a.o_out + row0 * R, voff128, 512u, rows_valid, lane);
        store_half<PITCH>(o[2], o[3], stage, a.o_out + row0 * R + 64, voff128, 512u, rows_valid, lane);
      }
      if constexpr (RESMODE == 1) {
#pragma unroll
        for (int j = 0; j < C::R32; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            o[j][4 * rq + 0] += xr[j][rq].x; o[j][4 * rq + 1] += xr[j][rq].y;
            o[j][4 * rq + 2] += xr[j][rq].z; o[j][4 * rq + 3] += xr[j][rq].w;
          }
      }
      if (tin) {
#pragma unroll
        for (int j = 0; j < C::R32; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            wmax = wn_absmax_acc(wmax, o[j][4 * rq + 0], o[j][4 * rq + 1], o[j][4 * rq + 2], o[j][4 * rq + 3]);
      }
      store_half<PITCH>(o[0], o[1], stage, a.x_out + row0 * R, voff128, 512u, rows_valid, lane);
      store_half<PITCH>(o[2], o[3], stage, a.x_out + row0 * R + 64, voff128, 512u, rows_valid, lane);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead chunks land before the LDS is given back
  if (a.absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) wn_absmax_publish_any(a.absmax_out, wmax);
  }
}

}  // namespace

int wn_layer_fwd_s128_supported(int R, int D, int KS) { return R == 128 && D == 128 && KS == 2; }

int wn_launch_layer_fwd_s128(const WnLayerFwdArgs& a, hipStream_t s) {
  if (!wn_layer_fwd_s128_supported(a.R, a.D, a.KS)) {
    wn_set_error("layer_fwd_s128: unsupported shape R=%d D=%d KS=%d", a.R, a.D, a.KS);
    return WN_E_UNSUPPORTED;
  }
  if ((int64_t)a.B * a.T * a.R * 4 >= (int64_t)1 << 32) { wn_set_error("layer_fwd_s128: activations beyond 4 GiB"); return WN_E_UNSUPPORTED; }
  if (a.z_out && (a.ldz % 4 != 0)) { wn_set_error("layer_fwd_s128: z row stride must be a multiple of 4"); return WN_E_UNSUPPORTED; }
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  if (tiles <= 0) return WN_OK;
  int64_t gx = (tiles + 7) / 8;
  if (gx > 256) gx = 256;                    // one persistent workgroup per CU
  const int resmode = a.residual ? 1 : 0;
  const bool save = a.ag_out != nullptr;
#define WN_S128_LAUNCH(RM_, SV_) \
  hipLaunchKernelGGL((wn_layer_fwd_s128_kernel<2, RM_, SV_>), dim3((unsigned)gx), dim3(512), 0, s, a)
  if (resmode == 0) { if (save) WN_S128_LAUNCH(0, true); else WN_S128_LAUNCH(0, false); }
  else { if (save) WN_S128_LAUNCH(1, true); else WN_S128_LAUNCH(1, false); }
#undef WN_S128_LAUNCH
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
