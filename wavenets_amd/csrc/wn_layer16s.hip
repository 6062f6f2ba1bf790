// Fused WaveNet residual block forward for blocks whose weights do not fit LDS (R = D = 128), split-precision MFMA,
// STREAMED weights (gfx950).   Reference: WaveNetLayer.call, src/layers.py:178-224 (depth-1 dilated stack).
//
// Same chain and register orientation as wn_layer16.hip -- time on lanes, u = b_d + sum_tap W_tap^T x[t - shift] ->
// z = tanh * sigmoid -> o = b_r + W_r^T z -> x_out = o + x, every product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
// v_mfma_f32_32x32x16_f16 -- but the fp16 hi|lo images of one block are 256 KiB (gated conv) + 64 KiB (1x1), twice the LDS.
// So a workgroup (4 waves = 128 rows per pass, two workgroups per CU) STREAMS them: the images are one sequence of 20
// chunks of 16 KiB (16 k-steps of the conv with all 8 row tiles, then 4 x two k-steps of the 1x1) that cycles through a
// 3-deep LDS ring filled by LDS-DMA, two chunks in flight, one raw barrier per chunk; the stream never stops at a tile
// boundary (the weights do not depend on the tile).  The activations of a k-step come by LDS-DMA too (two k-steps in
// flight, into the wave's otherwise idle output stage), so the only waits in the loop are counted s_waitcnt vmcnt(N).  u (32 x 256) stays
// in 128 accumulator registers per lane, the gate runs in registers and z feeds the 1x1 as the B operand as it stands: u
// never touches HBM and z is written once (for the folded skip contraction and backward), never read back here.
//
// Measured skeleton (tools/stream_probe.hip: the same DMA / barrier / store structure without the arithmetic): the L2-resident
// weight stream costs nothing beside the HBM streams (48 B/clk/CU alone, +3 % beside them) and the structure runs at the
// HBM rate of its activation traffic (5.2 TB/s).
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#ifdef WN_S128_DIAG
// phase stamps of the timing build (DIAG & 64): [workgroup][wave][pass 0..1][phase 0..9], read by wn_debug_s128_ts
__device__ unsigned long long wn_s128_ts[512 * 4 * 2 * 10];
#endif

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
  static constexpr int CHUNK = 16384, NBUF = 3, XB = 3, XBUF = 2048;
  static constexpr int WAVES = 4, THREADS = 256;
  static constexpr int PITCH = 36, STAGE = 32 * PITCH * 4;          // 4608: one 32 x 32 tile
  static constexpr int REGION = XB * XBUF;                          // per wave: activation ring, reused as the output stage
  static constexpr int BIAS = (2 * D + R) * 4;
  static constexpr int LDS = NBUF * CHUNK + WAVES * REGION + BIAS;  // 75264: two workgroups per CU
  static constexpr int PT = CHUNK / 16 / THREADS;      // weight-DMA instructions per thread and chunk (4)
  static constexpr int PX = 2;                         // activation-DMA instructions per lane and k-step
  static_assert(STAGE <= REGION, "the output stage lives in the activation ring");
};

// one 32 x 32 D-layout accumulator tile -> wave-private LDS stage -> 128-byte row segments in HBM.
// dst = wave-uniform address of the tile's first row (+ column offset), voff = this lane's byte offset inside a group of
// eight rows: the stores are scalar base + 32-bit lane offset.  The base goes through an empty asm so that it stays ONE
// scalar: otherwise hipcc re-associates (tensor + lane offset) + row, hoists that 64-bit VGPR pair of every output tensor
// out of the tile loop, spills it, and reloads it with s_waitcnt vmcnt(0) in the middle of the stores.
// FULL: all 32 rows exist (no per-row predicate, no branches).
template <int PITCH, bool FULL, bool NOSTORE = false, bool ADD = false>
__device__ __forceinline__ void store_tile(const f32x16& v, float* stage, float* dst, unsigned voff, unsigned ld_bytes,
                                           int rows_valid, int lane, const f32x4* add = nullptr) {
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
  // (the asm drops the address space: restore it, or the stores become flat_store)
  __attribute__((address_space(1))) char* base = (__attribute__((address_space(1))) char*)base0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 o = *reinterpret_cast<const f32x4*>(rd + i * 8 * PITCH);
    if constexpr (ADD) { o.x += add[i].x; o.y += add[i].y; o.z += add[i].z; o.w += add[i].w; }
    if (NOSTORE ? (o.x == 1.2345e-30f) : (FULL || i * 8 + (lane >> 3) < rows_valid))      // (NOSTORE: timing ablation)
      *(__attribute__((address_space(1))) f32x4*)(base + (uint64_t)((unsigned)(i * 8) * ld_bytes) + voff) = o;
  }
  asm volatile("" ::: "memory");
}

}  // namespace

// RESMODE 0: no residual; 1: x_out = o + (a.res ? a.res : a.x).  SAVE: the sigmoid is written for backward (training).
//
// Two workgroups of FOUR waves per CU (75 KiB of LDS, <= 256 registers per lane each): a wave that stores and then waits
// for a load waits for its stores too -- vmcnt is one in-order counter per wave -- so inside one workgroup the phases
// load | gate | store | 1x1 | store run one after the other (first form of this kernel, 8 waves per CU: 100 us per launch
// against 55 us for its bytes).  With two independent workgroups per CU one drains its stores while the other runs its
// products.
// DIAG != 0: timing ablations (knob 29; results are meaningless): 1 no global stores, 2 no activation requests, 4 no
// products, 8 no gate transcendentals, 16 no residual loads, 32 no weight requests
template <int KS, int RESMODE_, bool SAVE, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void wn_layer_fwd_s128_kernel(WnLayerFwdArgs a) {
  constexpr int RESMODE = (DIAG & 16) ? 0 : RESMODE_;
  constexpr bool NOST = (DIAG & 1) != 0;
  using C = G<KS>;
  constexpr int R = C::R, D = C::D, NC1 = C::NC1, NCH = C::NCH, PITCH = C::PITCH, PT = C::PT, PX = C::PX;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  const int tid = threadIdx.x;
  // the wave index through an SGPR: every LDS-DMA destination (M0) and every ring address is then scalar arithmetic
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  unsigned char* const xbuf = smem + C::NBUF * C::CHUNK + wave * C::REGION;
  float* stage = reinterpret_cast<float*>(xbuf);
  const unsigned smem_addr = lds_addr_of(smem), xbuf_addr = lds_addr_of(xbuf);
  float* sbias = reinterpret_cast<float*>(smem + C::NBUF * C::CHUNK + C::WAVES * C::REGION);
  for (int i = tid; i < 2 * D; i += C::THREADS) sbias[i] = a.bias_d[i];
  for (int i = tid; i < R; i += C::THREADS) sbias[2 * D + i] = a.bias_r[i];
  const float* lbias_d = sbias;
  const float* lbias_r = sbias + 2 * D;

  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  const int64_t per_pass = (int64_t)gridDim.x * C::WAVES;
  const int passes = (int)((ntiles + per_pass - 1) / per_pass);

  // weight chunk cc of the tile's sequence -> ring slot; piece i of a thread: 1 KiB per wave instruction
  const unsigned woff = (unsigned)tid * 16u;
  auto wpiece = [&](int cc, int slot, int i) {         // cc, i compile-time at every call; slot scalar
    const char* base = cc < NC1 ? reinterpret_cast<const char*>(a.frag_d) + (int64_t)cc * C::CHUNK
                                : reinterpret_cast<const char*>(a.frag_r) + (int64_t)(cc - NC1) * C::CHUNK;
    if constexpr (!(DIAG & 32)) dma16(base + 4096 * i, woff, smem_addr + slot * C::CHUNK + (C::THREADS * i + wave * 64) * 16);
  };
  auto next_slot = [](int s) { return s + 1 == C::NBUF ? 0 : s + 1; };

#ifdef WN_S128_DIAG
#define S128_TS(k) do { if constexpr ((DIAG & 64) != 0) { if (lane == 0 && pass < 2 && blockIdx.x < 512) \
  wn_s128_ts[((blockIdx.x * 4 + wave) * 2 + pass) * 10 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define S128_TS(k) do { } while (0)
#endif
  float wmax = 0.f;
  __syncthreads();                                     // bias table
  // the first two chunks of the stream
#pragma unroll
  for (int i = 0; i < PT; ++i) wpiece(0, 0, i);
#pragma unroll
  for (int i = 0; i < PT; ++i) wpiece(1, 1, i);
  int slot0 = 0;                                       // ring slot of the current tile's chunk 0 (scalar)

  for (int pass = 0; pass < passes; ++pass) {
    const int64_t tile = ((int64_t)pass * gridDim.x + blockIdx.x) * C::WAVES + wave;
    const bool live = tile < ntiles;                   // dead waves still take part in the barriers and the weight stream
    const int b = live ? (int)(tile / tiles_per_b) : 0;
    const int t0 = live ? (int)(tile % tiles_per_b) * 32 : 0;
    const int t = t0 + tl;
    const bool tin = live && t < a.T;
    const int rows_valid = live ? min(32, a.T - t0) : 0;
    const int64_t row0 = (int64_t)b * a.T + t0;

    // per-tap source row of this lane (clamped; masked rows are zeroed at use): 32-bit byte offsets from a scalar base
    // (the launcher checks that the tensors stay below 4 GiB)
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
      const unsigned dst = xbuf_addr + (c % C::XB) * C::XBUF;       // wave-uniform
      if constexpr (!(DIAG & 2)) {
        dma16(base, xoff[tap], dst);
        dma16(base + 32, xoff[tap], dst + 1024);
      }
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

    // Issue order per tile:  x0 x1 | step c: wait, barrier, then w(c+2) x(c+2) spread between the products of chunk c.
    // (w(0), w(1) of THIS tile were requested during the previous tile's last two steps, or before the loop.)  Requests
    // for chunk c + 2 go into the ring slot chunk c - 1 used, hence after the barrier every wave reaches once it has
    // finished chunk c - 1.  vmcnt retires loads in order, so "at most N outstanding" with N = the LOADS issued after
    // the ones chunk c needs means those have landed; stores in between only make the wait conservative.
    S128_TS(0);
    xdma(0);
    xdma(1);
    int slot = slot0;                                  // ring slot of chunk c, advanced per step (scalar)
    // =================== dilated causal conv: NC1 chunks ===================
    wn_static_for<NC1>([&](auto cc_) {
      constexpr int c = decltype(cc_)::value;
      // loads issued after x(c) [c <= 1: the prologue's] or after w(c), x(c) [step c - 2]:  w(c+1) x(c+1)
      constexpr int nyl = c == 0 ? PX : PT + (c + 1 < NC1 ? PX : 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nyl) : "memory");
      asm volatile("s_barrier" ::: "memory");
      if constexpr (c == 0) S128_TS(1);
      if constexpr (c == 8) S128_TS(2);
      const int slot2 = next_slot(next_slot(slot));
      const h8* wl = reinterpret_cast<const h8*>(smem + slot * C::CHUNK) + lane;
      const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + (c % C::XB) * C::XBUF) + lane;
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
        if constexpr (!(DIAG & 4)) {
          u[j] = mfma16(fr[j & 1][1], bh, u[j]);
          u[j] = mfma16(fr[j & 1][0], bl, u[j]);
          u[j] = mfma16(fr[j & 1][0], bh, u[j]);
        } else {
          u[j][0] += (float)fr[j & 1][1][0] * (float)bh[0] + (float)fr[j & 1][0][1] * (float)bl[1];
        }
        // one request of the look-ahead chunk per product block (a request costs the wave ~100 clocks of issue)
        if constexpr (j < PT) wpiece((c + 2) % NCH, slot2, j);
        if constexpr (j == PT && c + 2 < NC1) xdma(c + 2);
        __builtin_amdgcn_sched_barrier(0);
      });
      slot = next_slot(slot);
    });

    S128_TS(3);
    // =================== gate (in place): u[j] -> z, u[j + 4] -> sigmoid ===================
#pragma unroll
    for (int j = 0; j < C::D32; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sg = (DIAG & 8) ? u[j + C::D32][r] * 0.5f : wn_sigmoid_fast(u[j + C::D32][r]);
        u[j + C::D32][r] = sg;
        u[j][r] = ((DIAG & 8) ? u[j][r] : wn_tanh_fast(u[j][r])) * sg;
      }
    S128_TS(4);
    // lane offset inside a group of eight rows (row stride 512 B for the R / D wide tensors, ldz * 4 for z)
    const unsigned voff128 = (unsigned)(lane >> 3) * 512u + (unsigned)(lane & 7) * 16u;
    auto store_sig = [&](auto full_) {
      constexpr bool FULL = decltype(full_)::value;
      if constexpr (SAVE) {
#pragma unroll
        for (int j = 0; j < C::D32; ++j)
          store_tile<PITCH, FULL, NOST>(u[C::D32 + j], stage, a.ag_out + row0 * D + 32 * j, voff128, 512u, rows_valid, lane);
      }
    };
    auto store_z = [&](auto full_) {
      constexpr bool FULL = decltype(full_)::value;
      if (a.z_out) {
        const unsigned ldzb = (unsigned)a.ldz * 4u;
        const unsigned voffz = (unsigned)(lane >> 3) * ldzb + (unsigned)(lane & 7) * 16u;
#pragma unroll
        for (int j = 0; j < C::D32; ++j)
          store_tile<PITCH, FULL, NOST>(u[j], stage, a.z_out + row0 * a.ldz + 32 * j, voffz, ldzb, rows_valid, lane);
      }
    };
    if (rows_valid == 32) { store_sig(std::true_type{}); store_z(std::true_type{}); }
    else if (rows_valid > 0) { store_sig(std::false_type{}); store_z(std::false_type{}); }
    S128_TS(5);
    // The residual (x itself, or a separate tensor: dropout feeds the conv a dropped copy, queued generation ring rows) is
    // re-read here, ahead of the 1x1 whose products hide part of its latency, in the STORE layout of x_out (lane = 16 bytes of a
    // 128-byte row segment, 8 rows per instruction: whole cache lines) and added on the way out.  Keeping the newest tap's
    // 64 registers alive through the conv does not fit beside the 128 accumulators (the compiler spills all of them), and
    // a re-read in D layout touches every line from four instructions (measured: 17 of 93 us per launch).
    f32x4 xr[RESMODE == 1 ? C::R32 : 1][4];
    constexpr int NRES = RESMODE == 1 ? C::R32 * 4 : 0;  // ordinary loads the counted waits below have to step over
    if constexpr (RESMODE == 1) {
      const char* rbase0 = reinterpret_cast<const char*>(a.res ? a.res : a.x) + row0 * (R * 4);
      asm volatile("" : "+s"(rbase0));                   // one scalar base (see store_tile)
      const __attribute__((address_space(1))) char* rbase = (const __attribute__((address_space(1))) char*)rbase0;
#pragma unroll
      for (int j = 0; j < C::R32; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // rows past the end of the utterance read row 0 of the tile (never stored)
          const unsigned rr = (unsigned)(i * 8 + (lane >> 3));
          const unsigned roff = ((int)rr < rows_valid ? rr : 0u) * (R * 4u) + (unsigned)(lane & 7) * 16u;
          xr[j][i] = *(const __attribute__((address_space(1))) f32x4*)(rbase + 128 * j + roff);
        }
    }
    S128_TS(6);
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
      // loads younger than w(c): w(c+1), and the residual loads when they were issued after w(c) (c = NC1: w(c) from step
      // NC1 - 2, w(c+1) from step NC1 - 1, then the residual loads; c = NC1 + 1: the residual loads, then w(c+1))
      constexpr int nyl = PT + (cc < 2 ? NRES : 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nyl) : "memory");
      asm volatile("s_barrier" ::: "memory");
      if constexpr (cc == 0) S128_TS(7);
      const int slot2 = next_slot(next_slot(slot));
      const h8* wl = reinterpret_cast<const h8*>(smem + slot * C::CHUNK) + lane;
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
          if constexpr (!(DIAG & 4)) {
            o[j] = mfma16(fr[blk & 1][1], bh, o[j]);
            o[j] = mfma16(fr[blk & 1][0], bl, o[j]);
            o[j] = mfma16(fr[blk & 1][0], bh, o[j]);
          } else {
            o[j][0] += (float)fr[blk & 1][1][0] * (float)bh[0] + (float)fr[blk & 1][0][1] * (float)bl[1];
          }
          // (past the tile's end: the next tile's first chunks; on the last pass harmless re-reads nobody uses)
          if constexpr (blk < PT) wpiece((c + 2) % NCH, slot2, blk);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      slot = next_slot(slot);
    });
    slot0 = slot;                                      // NCH % NBUF != 0: the next tile starts where this one ended
    S128_TS(8);

    // =================== residual, range guard, x_out ===================
    auto store_out = [&](auto full_) {
      constexpr bool FULL = decltype(full_)::value;
      if (a.o_out) {
#pragma unroll
        for (int j = 0; j < C::R32; ++j)
          store_tile<PITCH, FULL, NOST>(o[j], stage, a.o_out + row0 * R + 32 * j, voff128, 512u, rows_valid, lane);
      }
      // range guard on max(|o|, |residual|) -- x_out = o + residual is formed on the way out, |x_out| <= 2 max(...): the
      // limit (3e4) keeps a factor 2 below the fp16 range
      if (tin) {
#pragma unroll
        for (int j = 0; j < C::R32; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            wmax = wn_absmax_acc(wmax, o[j][4 * rq + 0], o[j][4 * rq + 1], o[j][4 * rq + 2], o[j][4 * rq + 3]);
            if constexpr (RESMODE == 1) wmax = wn_absmax_acc(wmax, xr[j][rq].x, xr[j][rq].y, xr[j][rq].z, xr[j][rq].w);
          }
      }
#pragma unroll
      for (int j = 0; j < C::R32; ++j)
        store_tile<PITCH, FULL, NOST, RESMODE == 1>(o[j], stage, a.x_out + row0 * R + 32 * j, voff128, 512u, rows_valid, lane,
                                                    xr[RESMODE == 1 ? j : 0]);
    };
    if (rows_valid == 32) store_out(std::true_type{});
    else if (rows_valid > 0) store_out(std::false_type{});
    S128_TS(9);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead chunks land before the LDS is given back
  if (a.absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) wn_absmax_publish_any(a.absmax_out, wmax);
  }
}

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
  int64_t gx = (tiles + 3) / 4;
  if (gx > 512) gx = 512;                    // two persistent workgroups of four waves per CU
  const int resmode = a.residual ? 1 : 0;
  const bool save = a.ag_out != nullptr;
#define WN_S128_LAUNCH(RM_, SV_) \
  hipLaunchKernelGGL((wn_layer_fwd_s128_kernel<2, RM_, SV_>), dim3((unsigned)gx), dim3(256), 0, s, a)
#ifdef WN_S128_DIAG
  if (resmode == 1 && save && wn_debug_get(29) != 0) {   // timing ablations of the training form (tools/time_s128.py)
#define WN_S128_D(D_) case D_: hipLaunchKernelGGL((wn_layer_fwd_s128_kernel<2, 1, true, D_>), dim3((unsigned)gx), dim3(256), 0, s, a); break;
    switch (wn_debug_get(29)) {
      WN_S128_D(1) WN_S128_D(2) WN_S128_D(4) WN_S128_D(8) WN_S128_D(16) WN_S128_D(32) WN_S128_D(19) WN_S128_D(12) WN_S128_D(51) WN_S128_D(63) WN_S128_D(64)
      default: wn_set_error("layer_fwd_s128: no such ablation"); return WN_E_INVALID;
    }
#undef WN_S128_D
    WN_HIP_CHECK(hipGetLastError());
    return WN_OK;
  }
#endif
  if (resmode == 0) { if (save) WN_S128_LAUNCH(0, true); else WN_S128_LAUNCH(0, false); }
  else { if (save) WN_S128_LAUNCH(1, true); else WN_S128_LAUNCH(1, false); }
#undef WN_S128_LAUNCH
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

#ifdef WN_S128_DIAG
extern "C" int wn_debug_s128_ts(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(wn_s128_ts), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
