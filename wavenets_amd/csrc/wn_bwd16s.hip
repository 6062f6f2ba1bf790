// Backward-data chain for blocks whose weight images do not fit LDS (R = D = 128): two products per launch, STREAMED
// weights (gfx950, split-precision MFMA).  Same mathematics as wn_bwd_pair.hip (src/layers.py:199-223 reversed):
//   g_x(b+1)[t] = W_0(b+1) g_u(b+1)[t + d] + W_1(b+1) g_u(b+1)[t] + g_x(b+2)[t]
//   g_u(b)[t]   = gate'( W_r(b) g_x(b+1)[t] + V(b) dL/da[t] )                  (V(b) = W_s(b) W_f0: folded skip path)
// and the same key fact -- the g_x tile a wave has just produced is, unchanged, the B operand of the second product -- but
// the images are 256 KiB (A[128][512], reversed conv of block b+1) + 128 KiB ([W_r | V] of block b), so they are streamed
// like wn_layer16s.hip streams the forward's: 24 chunks of 16 KiB per tile (two k-steps x four row tiles each) through a
// 3-deep LDS ring filled by LDS-DMA, one raw barrier per chunk, two workgroups of four waves per CU.  The activations of a
// k-step (g_u(b+1) rows for the first product, dL/da rows for the second half of the second) come by LDS-DMA into a
// 4-slot ring per wave, four k-steps ahead; the ring doubles as the output stage.  The residual-path gradient and the
// gate's saved operands are ordinary loads issued where their registers are free; the counted waits step over them.
//
// Replaces two launches of the streamed rows GEMM per block boundary (128 + 110 us, profiles/r03_cfg4_*): g_x(b+1) is not
// re-read and one launch's ramp disappears.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split8s(const f32x4& q0, const f32x4& q1, float s, h8& hi, h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)(v[e] * s);
    hi[e] = h;
    lo[e] = (_Float16)__builtin_fmaf(v[e], s, -(float)h);
  }
}
// (see wn_layer16s.hip: scalar base + 32-bit lane offset, M0 = LDS address; inline assembly keeps the addresses scalar)
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ f32x4 ldg4(const __attribute__((address_space(1))) char* p) {
  return *(const __attribute__((address_space(1))) f32x4*)p;
}
__device__ __forceinline__ void pow2_scale(float m, float& sc, float& inv) {
  sc = 1.0f;
  inv = 1.0f;
  if (m > 0.f && m < 3.0e38f) {
    int e;
    (void)frexpf(m, &e);
    e = max(-100, min(100, e));
    sc = ldexpf(1.0f, -e);
    inv = ldexpf(1.0f, e);
  }
}

constexpr int R = 128, D = 128, F0 = 128;
constexpr int NC1 = 16;                       // chunks of the reversed conv: K = KS * 2D = 512 = 32 k-steps, two per chunk
constexpr int NC2 = 8;                        // chunks of [W_r | V]: K = R + F0 = 256 = 16 k-steps
constexpr int NCH = NC1 + NC2;                // 24 (a multiple of the ring depth: chunk c lives in slot c % 3)
constexpr int CHUNK = 16384, NBUF = 3, XB = 4, XBUF = 2048;
constexpr int WAVES = 4, THREADS = 256;
constexpr int PITCH = 36, STAGE = 32 * PITCH * 4;
constexpr int REGION = XB * XBUF;             // 8 KiB per wave: activation ring, reused as the output stage
constexpr int LDS = NBUF * CHUNK + WAVES * REGION;   // 81920: two workgroups per CU use the whole LDS
constexpr int PT = CHUNK / 16 / THREADS;      // 4 weight requests per thread and chunk
constexpr int PX = 2;                         // requests per lane and activation k-step
static_assert(NCH % NBUF == 0 && STAGE <= REGION, "ring / stage geometry");

// one 32 x 32 D-layout tile -> wave-private LDS stage -> 128-byte row segments (scalar base + lane offset, see wn_layer16s.hip)
template <bool FULL>
__device__ __forceinline__ void store_tile(const f32x16& v, float* stage, float* dst, unsigned voff, unsigned ld_bytes,
                                           int rows_valid, int lane) {
  const int tl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 o;
    o.x = v[4 * rq + 0]; o.y = v[4 * rq + 1]; o.z = v[4 * rq + 2]; o.w = v[4 * rq + 3];
    *reinterpret_cast<f32x4*>(stage + tl * PITCH + 8 * rq + 4 * h) = o;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float* rd = stage + (lane >> 3) * PITCH + (lane & 7) * 4;
  char* base0 = reinterpret_cast<char*>(dst);
  asm volatile("" : "+s"(base0));
  __attribute__((address_space(1))) char* base = (__attribute__((address_space(1))) char*)base0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(rd + i * 8 * PITCH);
    if (FULL || i * 8 + (lane >> 3) < rows_valid)
      *(__attribute__((address_space(1))) f32x4*)(base + (uint64_t)((unsigned)(i * 8) * ld_bytes) + voff) = o;
  }
  asm volatile("" ::: "memory");
}

}  // namespace

__global__ __launch_bounds__(256, 2) void wn_bwd_s128_kernel(WnBwdPairArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  unsigned char* const xbuf = smem + NBUF * CHUNK + wave * REGION;
  float* stage = reinterpret_cast<float*>(xbuf);
  const unsigned smem_addr = lds_addr_of(smem), xbuf_addr = lds_addr_of(xbuf);

  float sc1, inv1;
  pow2_scale(a.am_gu_in ? *a.am_gu_in : 0.f, sc1, inv1);
  const float gfmax = a.am_gf ? *a.am_gf : 0.f;

  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  const int64_t per_pass = (int64_t)gridDim.x * WAVES;
  const int passes = (int)((ntiles + per_pass - 1) / per_pass);

  const unsigned woff = (unsigned)tid * 16u;
  auto wpiece = [&](int cc, int i) {                   // piece i of chunk cc of the tile's sequence -> ring slot cc % NBUF
    const char* base = cc < NC1 ? reinterpret_cast<const char*>(a.wx16) + (int64_t)cc * CHUNK
                                : reinterpret_cast<const char*>(a.wu16) + (int64_t)(cc - NC1) * CHUNK;
    dma16(base + 4096 * i, woff, smem_addr + (cc % NBUF) * CHUNK + (THREADS * i + wave * 64) * 16);
  };

  float wmax_x = 0.f, wmax_u = 0.f;
#pragma unroll
  for (int i = 0; i < PT; ++i) wpiece(0, i);
#pragma unroll
  for (int i = 0; i < PT; ++i) wpiece(1, i);

  for (int pass = 0; pass < passes; ++pass) {
    const int64_t tile = ((int64_t)pass * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool live = tile < ntiles;                   // dead waves still take part in the barriers and the weight stream
    const int b = live ? (int)(tile / tiles_per_b) : 0;
    const int t0 = live ? (int)(tile % tiles_per_b) * 32 : 0;
    const int t = t0 + tl;
    const bool tin = live && t < a.T;
    const int rows_valid = live ? min(32, a.T - t0) : 0;
    const int64_t row0 = (int64_t)b * a.T + t0;
    const int64_t rowl = row0 + (tin ? tl : 0);        // this lane's row (rows past the end read the tile's first row)

    // ---- activation k-steps: A-phase k-step ks (0..31): tap = ks / 16 reads g_u(b+1) rows t + (tap == 0 ? d : 0), channels
    //      16 (ks % 16) ..; B-phase memory k-step f (0..7): dL/da row t, channels 16 f .. ----
    const bool ok0 = tin && t + a.dil < a.T, ok1 = tin;
    const unsigned goff0 = (unsigned)((row0 + (ok0 ? tl + a.dil : 0)) * 256 + 4 * h) * 4u;
    const unsigned goff1 = (unsigned)(rowl * 256 + 4 * h) * 4u;
    const unsigned foff = (unsigned)(rowl * F0 + 4 * h) * 4u;
    auto xdma = [&](int ks) {                          // -> ring slot ks % XB
      const int tap = ks / 16, kk = ks % 16;
      const char* base = reinterpret_cast<const char*>(a.gu_in) + 64 * kk;
      const unsigned dst = xbuf_addr + (ks % XB) * XBUF;
      dma16(base, tap == 0 ? goff0 : goff1, dst);
      dma16(base + 32, tap == 0 ? goff0 : goff1, dst + 1024);
    };
    auto fdma = [&](int f) {                           // -> ring slot f % XB
      const char* base = reinterpret_cast<const char*>(a.gf) + 64 * f;
      const unsigned dst = xbuf_addr + (f % XB) * XBUF;
      dma16(base, foff, dst);
      dma16(base + 32, foff, dst + 1024);
    };

    // ---- residual-path gradient g_x(b+2) of this tile (D layout): ordinary loads, older than every request of the tile ----
    f32x4 addc[4][4];
    {
      const char* r0_ = reinterpret_cast<const char*>(a.gx_res);
      asm volatile("" : "+s"(r0_));
      const __attribute__((address_space(1))) char* rb = (const __attribute__((address_space(1))) char*)r0_;
      const unsigned ro = (unsigned)(rowl * R + 4 * h) * 4u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) addc[j][rq] = ldg4(rb + (128 * j + 32 * rq) + ro);
    }
    xdma(0); xdma(1); xdma(2); xdma(3);

    // =================== g_x(b+1) = reversed dilated conv of g_u(b+1): NC1 chunks ===================
    // Issue order:  x0..x3 | step c: w(c+2), then x(2c+4) x(2c+5) once this step's two buffers have been read.
    // vmcnt retires loads in order: "at most N outstanding" with N = the LOADS issued after the youngest one step c needs
    // (x(2c+1), requested at the end of step c - 2) means everything it needs has landed.
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    wn_static_for<NC1>([&](auto cc_) {
      constexpr int c = decltype(cc_)::value;
      // younger than x(2c+1):  c == 0: x2 x3 | c >= 1: w(c+1) and, while they exist, x(2c+2) x(2c+3)
      constexpr int nyl = c == 0 ? 2 * PX : PT + (2 * c + 3 < 2 * NC1 ? 2 * PX : 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nyl) : "memory");
      asm volatile("s_barrier" ::: "memory");
      const h8* wl = reinterpret_cast<const h8*>(smem + (c % NBUF) * CHUNK) + lane;
      h8 fr[2][2];
      fr[0][0] = wl[0];
      fr[0][1] = wl[64];
      wn_static_for<2>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int ks = 2 * c + k;
        const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + (ks % XB) * XBUF) + lane;
        const f32x4 q0 = xl[0], q1 = xl[64];
        h8 bh, bl;
        split8s(q0, q1, (ks / 16 == 0 ? ok0 : ok1) ? sc1 : 0.f, bh, bl);   // masked rows contribute zero
        wn_static_for<4>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int blk = k * 4 + j;
          if constexpr (blk + 1 < 8) {
            fr[(blk + 1) & 1][0] = wl[((blk + 1) * 2 + 0) * 64];
            fr[(blk + 1) & 1][1] = wl[((blk + 1) * 2 + 1) * 64];
          }
          acc[j] = mfma16(fr[blk & 1][1], bh, acc[j]);
          acc[j] = mfma16(fr[blk & 1][0], bl, acc[j]);
          acc[j] = mfma16(fr[blk & 1][0], bh, acc[j]);
          // requests of this step, one per product block: the look-ahead weight chunk first, then (both activation
          // buffers of the step have been read by now) the k-steps two steps ahead
          if constexpr (blk < PT) wpiece((c + 2) % NCH, blk);
          if constexpr (blk == 4 && 2 * c + 4 < 2 * NC1) xdma(2 * c + 4);
          if constexpr (blk == 5 && 2 * c + 5 < 2 * NC1) xdma(2 * c + 5);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    });

    // ---- epilogue 1: + g_x(b+2); rows past the utterance are exact zeros (they feed the next product) ----
    f32x16 gx[4];
    float tmax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float cv[4] = {addc[j][rq].x, addc[j][rq].y, addc[j][rq].z, addc[j][rq].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = tin ? acc[j][4 * rq + e] * inv1 + cv[e] : 0.f;
          gx[j][4 * rq + e] = v;
          tmax = fmaxf(tmax, fabsf(v));
        }
      }
    const unsigned voff128 = (unsigned)(lane >> 3) * 512u + (unsigned)(lane & 7) * 16u;
    if (rows_valid == 32) {
#pragma unroll
      for (int j = 0; j < 4; ++j) store_tile<true>(gx[j], stage, a.gx_out + row0 * R + 32 * j, voff128, 512u, rows_valid, lane);
    } else if (rows_valid > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) store_tile<false>(gx[j], stage, a.gx_out + row0 * R + 32 * j, voff128, 512u, rows_valid, lane);
    }
    wmax_x = fmaxf(wmax_x, tmax);
    // per-tile power-of-two scale of the second product's B operands (the tensor's max-abs is not known while it is produced)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    float sc2, inv2;
    pow2_scale(fmaxf(tmax, gfmax), sc2, inv2);

    // =================== g_z = W_r g_x (registers) + V dL/da (ring): NC2 chunks ===================
    // Issue order:  step 16: w18 f0 f1 | 17: w19 f2 f3 | 18: w20 | 19: w21 | [gate operands: 32 ordinary loads] |
    //               20: w22 f4 f5 | 21: w23 f6 f7 | 22: w0' | 23: w1'
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 sg[4][4], zz[4][4];
    constexpr int NGATE = 32;                           // ordinary loads between steps 19 and 20
    wn_static_for<NC2>([&](auto cc_) {
      constexpr int cc = decltype(cc_)::value;
      constexpr int c = NC1 + cc;
      if constexpr (cc == 4) {
        // gate-derivative operands of block b (D layout): their registers are free now (the g_x tiles are consumed)
        const char* s0_ = reinterpret_cast<const char*>(a.ag);
        const char* z0_ = reinterpret_cast<const char*>(a.z);
        asm volatile("" : "+s"(s0_), "+s"(z0_));
        const __attribute__((address_space(1))) char* sb = (const __attribute__((address_space(1))) char*)s0_;
        const __attribute__((address_space(1))) char* zb = (const __attribute__((address_space(1))) char*)z0_;
        const unsigned so = (unsigned)(rowl * D + 4 * h) * 4u;
        const unsigned zo = (unsigned)(rowl * a.ldz + 4 * h) * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            sg[j][rq] = ldg4(sb + (128 * j + 32 * rq) + so);
            zz[j][rq] = ldg4(zb + (128 * j + 32 * rq) + zo);
          }
      }
      // youngest load the step needs and what was issued after it:
      //   cc 0: w16 (step 14)                 <- w17                                   : PT
      //   cc 1: w17 (step 15)                 <- w18 f0 f1                             : PT + 2 PX
      //   cc 2: w18 (step 16; f0 f1 not yet)  <- f0 f1 w19 f2 f3                       : PT + 4 PX
      //   cc 3: w19 (step 17)                 <- f2 f3 w20                             : PT + 2 PX
      //   cc 4: w20 (step 18; needs f0 f1: older)  <- w21 + gate loads                 : PT + NGATE
      //   cc 5: w21 (step 19; f2 f3 older)    <- gate loads, w22 f4 f5                 : NGATE + PT + 2 PX
      //   cc 6: f5 (step 20; w22 older)       <- w23 f6 f7                             : PT + 2 PX
      //   cc 7: f7 (step 21; w23 older)       <- w0'                                   : PT
      constexpr int nyl = cc == 0 ? PT : cc == 1 ? PT + 2 * PX : cc == 2 ? PT + 4 * PX : cc == 3 ? PT + 2 * PX
                        : cc == 4 ? PT + NGATE : cc == 5 ? NGATE + PT + 2 * PX : cc == 6 ? PT + 2 * PX : PT;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nyl) : "memory");
      asm volatile("s_barrier" ::: "memory");
      const h8* wl = reinterpret_cast<const h8*>(smem + (c % NBUF) * CHUNK) + lane;
      h8 fr[2][2];
      fr[0][0] = wl[0];
      fr[0][1] = wl[64];
      wn_static_for<2>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int ks = 2 * cc + k;                   // k-step of the second product: 0..7 from g_x, 8..15 from dL/da
        f32x4 q0, q1;
        float s = sc2;
        if constexpr (ks < 8) {
          constexpr int jz = ks / 2, r0 = 8 * (ks % 2);
          q0.x = gx[jz][r0 + 0]; q0.y = gx[jz][r0 + 1]; q0.z = gx[jz][r0 + 2]; q0.w = gx[jz][r0 + 3];
          q1.x = gx[jz][r0 + 4]; q1.y = gx[jz][r0 + 5]; q1.z = gx[jz][r0 + 6]; q1.w = gx[jz][r0 + 7];
        } else {
          const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + ((ks - 8) % XB) * XBUF) + lane;
          q0 = xl[0]; q1 = xl[64];
          s = tin ? sc2 : 0.f;
        }
        h8 bh, bl;
        split8s(q0, q1, s, bh, bl);
        wn_static_for<4>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int blk = k * 4 + j;
          if constexpr (blk + 1 < 8) {
            fr[(blk + 1) & 1][0] = wl[((blk + 1) * 2 + 0) * 64];
            fr[(blk + 1) & 1][1] = wl[((blk + 1) * 2 + 1) * 64];
          }
          acc[j] = mfma16(fr[blk & 1][1], bh, acc[j]);
          acc[j] = mfma16(fr[blk & 1][0], bl, acc[j]);
          acc[j] = mfma16(fr[blk & 1][0], bh, acc[j]);
          // (past the tile's end: the next tile's first chunks; on the last pass harmless re-reads nobody uses)
          if constexpr (blk < PT) wpiece((c + 2) % NCH, blk);
          if constexpr (blk == 4 && (cc == 0 || cc == 1 || cc == 4 || cc == 5)) fdma(cc < 2 ? 2 * cc : 2 * cc - 4);
          if constexpr (blk == 5 && (cc == 0 || cc == 1 || cc == 4 || cc == 5)) fdma(cc < 2 ? 2 * cc + 1 : 2 * cc - 3);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    });

    // ---- epilogue 2: gate derivative (filter half -> columns 0..127, gate half -> columns 128..255 of g_u(b)), tile by
    //      tile so that a tile's accumulator and gate operands die as soon as both its halves are on their way ----
    const unsigned voff256 = (unsigned)(lane >> 3) * 1024u + (unsigned)(lane & 7) * 16u;
    float umax = 0.f;
    wn_static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      f32x16 of, og;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float gv[4] = {sg[j][rq].x, sg[j][rq].y, sg[j][rq].z, sg[j][rq].w};
        const float zv[4] = {zz[j][rq].x, zz[j][rq].y, zz[j][rq].z, zz[j][rq].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dz = acc[j][4 * rq + e] * inv2;
          const float vf = tin ? wn_gate_bwd_f(dz, gv[e], zv[e]) : 0.f;
          const float vg = tin ? wn_gate_bwd_g(dz, gv[e], zv[e]) : 0.f;
          of[4 * rq + e] = vf;
          og[4 * rq + e] = vg;
          umax = fmaxf(umax, fmaxf(fabsf(vf), fabsf(vg)));
        }
      }
      if (rows_valid == 32) {
        store_tile<true>(of, stage, a.gu_out + row0 * 256 + 32 * j, voff256, 1024u, rows_valid, lane);
        store_tile<true>(og, stage, a.gu_out + row0 * 256 + 128 + 32 * j, voff256, 1024u, rows_valid, lane);
      } else if (rows_valid > 0) {
        store_tile<false>(of, stage, a.gu_out + row0 * 256 + 32 * j, voff256, 1024u, rows_valid, lane);
        store_tile<false>(og, stage, a.gu_out + row0 * 256 + 128 + 32 * j, voff256, 1024u, rows_valid, lane);
      }
    });
    wmax_u = fmaxf(wmax_u, umax);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead chunks land before the LDS is given back
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    wmax_x = fmaxf(wmax_x, __shfl_xor(wmax_x, o));
    wmax_u = fmaxf(wmax_u, __shfl_xor(wmax_u, o));
  }
  if (lane == 0) {
    if (a.am_gx) wn_absmax_publish(a.am_gx, wmax_x);
    if (a.am_gu) wn_absmax_publish(a.am_gu, wmax_u);
  }
}

int wn_bwd_s128_supported(int R_, int D_, int KS, int F0_) { return (R_ == 128 && D_ == 128 && KS == 2 && F0_ == 128) ? 1 : 0; }

int wn_launch_bwd_s128(const WnBwdPairArgs& a, hipStream_t s) {
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  if (tiles <= 0) return WN_OK;
  if ((int64_t)a.B * a.T * 256 * 4 >= (int64_t)1 << 32) { wn_set_error("bwd_s128: activations beyond 4 GiB"); return WN_E_UNSUPPORTED; }
  if (a.ldz % 4 != 0) { wn_set_error("bwd_s128: z row stride must be a multiple of 4"); return WN_E_UNSUPPORTED; }
  int64_t gx = (tiles + 3) / 4;
  if (gx > 512) gx = 512;                    // two persistent workgroups of four waves per CU
  hipLaunchKernelGGL(wn_bwd_s128_kernel, dim3((unsigned)gx), dim3(256), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
