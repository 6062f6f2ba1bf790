// Rows GEMM, split-precision MFMA variant (gfx950):
//   Y[t][n] = epi( sum_seg sum_k X_seg[t - shift_seg][k] * W[k_seg + k][n] )
// Same contract as wn_gemm_rows_kernel (wn_gemm.hip) but each fp32 product is evaluated as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16 (see wn_layer16.hip), and the kernel
// is organised as a streaming GEMM:
//   * persistent workgroups of 8 waves (256 rows per pass); the fp16 hi|lo weight image of ALL
//     segments is one concatenated sequence of 2 KiB (k-step, row-tile) blocks that the
//     workgroup streams through a double-buffered LDS ring in chunks (one barrier per chunk),
//     so L2 weight traffic drops 8x versus one weight pass per wave
//   * output tiles leave through a wave-private LDS stage as full row segments
//   * gradient operands are pre-scaled by an exact power of two taken from a device-side
//     running max-abs of the tensor (written by the producing kernel), so that their fp16 lo
//     parts do not sink into the fp16 subnormal range; the accumulators are scaled back exactly.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 wn_mfma16g(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void wn_split8g(const f32x4& q0, const f32x4& q1, float s, h8& hi, h8& lo) {
  // hi = fp16(q * s); lo = fp16(q * s - hi) with the product unrounded (explicit fma: one operation
  // fewer in a loop that is VALU-bound on this split, and independent of the contraction mode)
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)(v[e] * s);
    hi[e] = h;
    lo[e] = (_Float16)__builtin_fmaf(v[e], s, -(float)h);
  }
}

// global-address-space 16-byte load.  Pointers that went through a per-segment select lose their
// address space and hipcc emits flat_load: flat loads also count on lgkmcnt, so every LDS fragment
// wait would drain the activation prefetch, and the compiler waits vmcnt(0) while one is pending.
__device__ __forceinline__ f32x4 wn_ldg4(const float* p) {
  return *(const __attribute__((address_space(1))) f32x4*)(p);
}

// chunk = CH (k-step, row-tile) blocks of 2 KiB  ->  KSC = CH / JT k-steps per chunk
template <int JT>
struct WnG16 {
  static constexpr int KSC = 2;                       // k-steps per chunk
  static constexpr int CH = KSC * JT;                 // (k-step, row-tile) blocks per chunk
  static constexpr int CHUNK_BYTES = CH * 2048;       // 8 / 16 KiB of weights for JT = 2 / 4
  static constexpr int XBUF_BYTES = KSC * 2 * 1024;   // one chunk of a wave's activations (two fit in its stage)
  static constexpr int SC = 64;                       // stage width in channels
  static constexpr int PITCH = SC + 4;
  static constexpr int STAGE_BYTES = 32 * PITCH * 4;  // 8704
  static constexpr int WAVES = 8;
  static constexpr int NBUF = 4;                      // weight ring: chunk c lives in buffer c % NBUF, two chunks in flight
  static constexpr int LDS_BYTES = NBUF * CHUNK_BYTES + WAVES * STAGE_BYTES;   // 135168
};

// epilogue shared by the streamed and the resident kernels: bias / row bias / residual / activation
// (or activation derivative, or gate derivative) in registers, then staged full-row stores
// row operands of the epilogue fetched ahead of time (resident kernel): addc, aux (saved activations,
// or tanh for the gate derivative), aux2 (sigmoid for the gate derivative), all in D layout
template <int JT>
struct WnG16Pre {
  f32x4 addc[JT][4], aux[JT][4], aux2[JT][4];
};

// EPI >= 0 fixes the epilogue kind at compile time (the resident kernels know theirs): the other kinds'
// code and their live registers disappear
template <int JT, int PITCH, int PRE = 0, int EPI = -1>
__device__ __forceinline__ void wn_g16_epilogue(const WnGemmArgs& a, f32x16 (&acc)[JT], float inv_sc, int jb, int b,
                                                int t, int64_t row0, int rows_valid, float* stage, int lane,
                                                float& wmax, const WnG16Pre<JT>* pre = nullptr) {
  const int tl = lane & 31, h = lane >> 5;
  if constexpr (EPI == WN_EPI_GATE_FWD) {
    // Forward gate of a residual block whose weights do not fit LDS (R = D = 128): the image's row tiles
    // are ordered [f f g g] per 128-column block, so this workgroup holds filter channels
    // c0 .. c0 + 63 (tiles 0, 1) and their gate channels (tiles 2, 3) and can gate locally.
    static_assert(JT == 4, "gate-forward epilogue needs [f f g g] column blocks");
    const int D = a.N / 2;
    const int c0 = 16 * jb;                              // jb = 4 * column block -> 64 output channels per block
    const bool tin = t < a.T;
    for (int part = 0; part < (a.y2 ? 2 : 1); ++part) {
      f32x16 outv[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int n = c0 + 32 * jj + 8 * rq + 4 * h;   // channel inside [0, D)
          float z[4] = {0.f, 0.f, 0.f, 0.f};
          if (tin) {
            const f32x4 bf = wn_ldg4(a.bias + n), bg = wn_ldg4(a.bias + D + n);
            const float bfv[4] = {bf.x, bf.y, bf.z, bf.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float f = acc[jj][4 * rq + e] * inv_sc + bfv[e];
              const float g = acc[jj + 2][4 * rq + e] * inv_sc + bgv[e];
              const float sg = wn_sigmoid_fast(g);
              z[e] = part == 0 ? wn_tanh_fast(f) * sg : sg;
            }
          }
          outv[jj][4 * rq + 0] = z[0]; outv[jj][4 * rq + 1] = z[1]; outv[jj][4 * rq + 2] = z[2]; outv[jj][4 * rq + 3] = z[3];
        }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 o;
          o.x = outv[jj][4 * rq + 0]; o.y = outv[jj][4 * rq + 1]; o.z = outv[jj][4 * rq + 2]; o.w = outv[jj][4 * rq + 3];
          *reinterpret_cast<f32x4*>(stage + tl * PITCH + 32 * jj + 8 * rq + 4 * h) = o;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float* dst = (part == 0 ? a.y + row0 * a.ldy : a.y2 + row0 * a.ld_y2) + c0;
      const int64_t ldd = part == 0 ? a.ldy : a.ld_y2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 4 + (lane >> 4);
        const int col = (lane & 15) * 4;
        const f32x4 o = *reinterpret_cast<const f32x4*>(stage + r * PITCH + col);
        if (r < rows_valid) *reinterpret_cast<f32x4*>(dst + (int64_t)r * ldd + col) = o;
      }
      asm volatile("" ::: "memory");
    }
    return;
  }
  {
    // ---- epilogue in registers (D layout), then staged row stores, 64 channels at a time ----
    const int64_t row = row0 + tl;
    const bool tin = t < a.T;
    // GATE_BWD produces two outputs per accumulator; they are staged one after the other
    const int epi = EPI >= 0 ? EPI : a.epi;
    const int nout = (epi == WN_EPI_GATE_BWD) ? 2 : 1;
    wn_static_for<(JT + 1) / 2>([&](auto gc) {
      constexpr int g2 = decltype(gc)::value;          // pair of row tiles -> 64 channels
      for (int part = 0; part < nout; ++part) {
        f32x16 outv[2];
        wn_static_for<2>([&](auto jc) {
          constexpr int jj = decltype(jc)::value;
          constexpr int j = 2 * g2 + jj;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int n0 = 32 * (jb + j) + 8 * rq + 4 * h;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (j < JT && n0 < a.N && tin) {
              v[0] = acc[j < JT ? j : 0][4 * rq + 0] * inv_sc; v[1] = acc[j < JT ? j : 0][4 * rq + 1] * inv_sc;
              v[2] = acc[j < JT ? j : 0][4 * rq + 2] * inv_sc; v[3] = acc[j < JT ? j : 0][4 * rq + 3] * inv_sc;
              if (a.bias) {
                const f32x4 bv = wn_ldg4(a.bias + n0);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
              }
              if (a.rowbias) {
                const f32x4 rb = wn_ldg4(a.rowbias + (int64_t)b * a.ld_rowbias + n0);
                v[0] += rb.x; v[1] += rb.y; v[2] += rb.z; v[3] += rb.w;
              }
              if (a.addc) {
                f32x4 cv;
                if constexpr (PRE == 1) cv = pre->addc[j < JT ? j : 0][rq];
                else cv = wn_ldg4(a.addc + row * a.ld_addc + n0);
                v[0] += cv.x; v[1] += cv.y; v[2] += cv.z; v[3] += cv.w;
              }
              if (epi == WN_EPI_PLAIN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = wn_act(v[e], a.act);
              } else if (epi == WN_EPI_DACT) {
                f32x4 yv;
                if constexpr (PRE == 3) yv = pre->aux[j < JT ? j : 0][rq];
                else yv = wn_ldg4(a.aux + row * a.ld_aux + n0);
                v[0] *= wn_dact_from_y(yv.x, a.act); v[1] *= wn_dact_from_y(yv.y, a.act);
                v[2] *= wn_dact_from_y(yv.z, a.act); v[3] *= wn_dact_from_y(yv.w, a.act);
              } else {
                f32x4 gv, zv;        // saved sigmoid, gated activation
                if constexpr (PRE == 2) {
                  gv = pre->aux[j < JT ? j : 0][rq];
                  zv = pre->aux2[j < JT ? j : 0][rq];
                } else {
                  gv = wn_ldg4(a.aux + row * a.ld_aux + n0);
                  zv = wn_ldg4(a.aux2 + row * a.ld_aux2 + n0);
                }
                // each pass evaluates only the derivative it stores (filter half, then gate half)
                if (part == 0) {
                  v[0] = wn_gate_bwd_f(v[0], gv.x, zv.x); v[1] = wn_gate_bwd_f(v[1], gv.y, zv.y);
                  v[2] = wn_gate_bwd_f(v[2], gv.z, zv.z); v[3] = wn_gate_bwd_f(v[3], gv.w, zv.w);
                } else {
                  v[0] = wn_gate_bwd_g(v[0], gv.x, zv.x); v[1] = wn_gate_bwd_g(v[1], gv.y, zv.y);
                  v[2] = wn_gate_bwd_g(v[2], gv.z, zv.z); v[3] = wn_gate_bwd_g(v[3], gv.w, zv.w);
                }
              }
              wmax = wn_absmax_acc(wmax, v[0], v[1], v[2], v[3]);
            }
            outv[jj][4 * rq + 0] = v[0]; outv[jj][4 * rq + 1] = v[1];
            outv[jj][4 * rq + 2] = v[2]; outv[jj][4 * rq + 3] = v[3];
          }
        });
        // stage (32 x 64) and store the valid part as row segments
        const int c0 = 32 * (jb + 2 * g2);             // first channel of this group
        if (c0 < a.N) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              f32x4 o;
              o.x = outv[jj][4 * rq + 0]; o.y = outv[jj][4 * rq + 1]; o.z = outv[jj][4 * rq + 2]; o.w = outv[jj][4 * rq + 3];
              *reinterpret_cast<f32x4*>(stage + tl * PITCH + 32 * jj + 8 * rq + 4 * h) = o;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const int ncols = min(min(64, 32 * (JT - 2 * g2)), a.N - c0);   // multiple of 32; never past this block's tiles
          float* dst = a.y + row0 * a.ldy + (part ? a.N : 0) + c0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int r = i * 4 + (lane >> 4);
            const int col = (lane & 15) * 4;
            const f32x4 o = *reinterpret_cast<const f32x4*>(stage + r * PITCH + col);
            if (r < rows_valid && col < ncols) *reinterpret_cast<f32x4*>(dst + (int64_t)r * a.ldy + col) = o;
          }
          asm volatile("" ::: "memory");
        }
      }
    });
  }
}

template <int JT, int EPI>
__global__ __launch_bounds__(512, 2) void wn_gemm_rows16_kernel(WnGemmArgs a, const float* w16, int nks_total,
                                                                const float* absmax_in0, const float* absmax_in1,
                                                                float* absmax_out) {
  using G = WnG16<JT>;
  constexpr int KSC = G::KSC, PITCH = G::PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  float* stage = reinterpret_cast<float*>(smem + G::NBUF * G::CHUNK_BYTES + wave * G::STAGE_BYTES);

  // exact power-of-two operand scale from the producers' running max-abs
  float sc = 1.0f, inv_sc = 1.0f;
  if (absmax_in0) {
    float m = *absmax_in0;
    if (absmax_in1) m = fmaxf(m, *absmax_in1);
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);               // m = f * 2^e, f in [0.5, 1)
      e = max(-100, min(100, e));
      sc = ldexpf(1.0f, -e);             // scaled values lie in [-1, 1)
      inv_sc = ldexpf(1.0f, e);
    }
  }

  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  const int nchunks = (nks_total + KSC - 1) / KSC;
  // 1-D grid of (row group, column block) pairs.  Workgroup ids that differ by 8 land on the same XCD
  // (round-robin dispatch), so the column blocks of one row group are placed 8 ids apart: they run
  // at the same time behind the same L2 and the rows are fetched from HBM once, not once per column block.
  const int ny = (a.JTtot + JT - 1) / JT;
  const int gxw = gridDim.x / ny;                      // workgroups per column block
  int xb, cb;
  if (ny > 1 && gxw % 8 == 0) {
    const int grp = blockIdx.x / (8 * ny), rem = blockIdx.x % (8 * ny);
    cb = rem / 8;
    xb = grp * 8 + rem % 8;
  } else {
    cb = blockIdx.x % ny;
    xb = blockIdx.x / ny;
  }
  const int64_t passes = (ntiles + (int64_t)gxw * G::WAVES - 1) / ((int64_t)gxw * G::WAVES);
  const int jb = cb * JT;                              // first row tile of this block (N > 32*JT)

  // segment boundaries in k-steps; the segment fields themselves are read from the kernel-argument
  // block by a wave-uniform index (see the resident kernel: local copies would live in scratch)
  static_assert(WN_MAXSEG == 4, "segment select below is written out for four segments");
  const int e0 = (a.seg[0].K + 15) >> 4;
  const int e1 = a.nseg > 1 ? e0 + ((a.seg[1].K + 15) >> 4) : nks_total;
  const int e2 = a.nseg > 2 ? e1 + ((a.seg[2].K + 15) >> 4) : nks_total;
  const int plane_ks0 = a.seg[0].plane_k > 0 ? a.seg[0].plane_k / 16 : 0;     // only segment 0 may be planar
  const int64_t plane_st0 = a.seg[0].plane_stride;
  float wmax = 0.f;
  unsigned char* const xbuf = reinterpret_cast<unsigned char*>(stage);          // 2 x XBUF_BYTES inside the wave's stage
  static_assert(2 * G::XBUF_BYTES <= G::STAGE_BYTES, "activation double buffer must fit the output stage");
  constexpr int PT = G::CHUNK_BYTES / 16 / 512;        // weight-DMA instructions per thread and chunk
  constexpr int PX = KSC * 2;                          // activation-DMA instructions per lane and chunk

  for (int64_t pass = 0; pass < passes; ++pass) {
    const int64_t tile = (pass * gxw + xb) * G::WAVES + wave;
    const bool live = tile < ntiles;                   // dead waves still take part in the barriers
    const int b = live ? (int)(tile / tiles_per_b) : 0;
    const int t0 = live ? (int)(tile % tiles_per_b) * 32 : 0;
    const int t = t0 + tl;
    const int rows_valid = live ? min(32, a.T - t0) : 0;
    const int64_t row0 = (int64_t)b * a.T + t0;

    // Everything the K loop consumes arrives by LDS-DMA: the weight chunk (shared, ring of NBUF buffers)
    // and this wave's own activations (two buffers inside its otherwise idle output stage; each lane
    // reads back the 16 bytes it requested).  With no register-destination load in the loop the only
    // waits are the counted s_waitcnt below: hipcc drains every LDS-DMA (vmcnt(0)) at the first use of
    // an ordinary load's result while one is in flight.
    // activations of chunk c -> activation buffer (c & 1); ok[k] = this lane's row is inside the
    // utterance for k-step k's segment (masked rows / k-steps read a clamped address and are zeroed at use)
    auto xdma = [&](int c, bool (&okv)[KSC]) {
#pragma unroll
      for (int k = 0; k < KSC; ++k) {
        const int ks_in = c * KSC + k;
        const int ks = min(ks_in, nks_total - 1);
        const int si = (ks >= e0) + (ks >= e1) + (ks >= e2);            // wave-uniform segment index
        const int kk = ks - (si == 0 ? 0 : si == 1 ? e0 : si == 2 ? e1 : e2);
        const float* bx = a.seg[si].x;
        const int ld = a.seg[si].ldx, sh = a.seg[si].shift;
        const int ts = t - sh;
        const bool ok = live && t < a.T && ts >= 0 && ts < a.T;
        const float* src = bx + ((int64_t)b * a.T + (ok ? ts : 0)) * ld + 4 * h + 16 * kk;
        if (plane_ks0 > 0 && si == 0) src += (int64_t)(kk / plane_ks0) * plane_st0 - (int64_t)(kk / plane_ks0) * plane_ks0 * 16;
        okv[k] = ok && ks_in < nks_total;
        unsigned char* dst = xbuf + ((c & 1) * KSC + k) * 2048;          // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8),
                                         (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
      }
    };
    // weight chunk c: global -> LDS buffer (c % NBUF): every wave instruction moves one contiguous KiB
    // (wave-uniform LDS base + lane * 16).  Blocks past the end of the image are clamped to a valid
    // block: their activations are zero, so they only need to be finite.
    auto wdma = [&](int c) {
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const int f = tid + 512 * i;                   // 16-byte piece inside the chunk
        const int blk = f >> 7, within = f & 127;
        int ks = c * KSC + blk / JT, j = jb + blk % JT;
        ks = min(ks, nks_total - 1);
        j = min(j, a.JTtot - 1);
        const f32x4* src = reinterpret_cast<const f32x4*>(w16) + ((int64_t)ks * a.JTtot + j) * 128 + within;
        unsigned char* dst = smem + (c % G::NBUF) * G::CHUNK_BYTES + (512 * i + wave * 64) * 16;   // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };

    f32x16 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Issue order  w(0), x(0), w(1) | x(1), w(2) | x(2), w(3) | ...   vmcnt retires in order, so "at most
    // N outstanding" with N = the requests issued after x(c) -- w(c+1), x(c+1), w(c+2) -- means x(c)
    // and w(c) have landed while the newer ones stay in flight.  One raw barrier per chunk (for the
    // other waves' weight pieces); a wave is at most one chunk ahead of the slowest, so buffer
    // (c + 2) % 4 is never one that is still being read.
    bool oka[KSC], okb[KSC];
    __syncthreads();                                   // previous pass finished with the ring and this stage
    wdma(0);
    xdma(0, oka);
    if (1 < nchunks) wdma(1);
    auto arrive = [&](int c) {
      if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT + PX) : "memory");
      else if (c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT + PX) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    };
    auto compute = [&](int c, const bool (&okv)[KSC]) {
      const h8* wl = reinterpret_cast<const h8*>(smem + (c % G::NBUF) * G::CHUNK_BYTES) + lane;
      const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + (c & 1) * G::XBUF_BYTES) + lane;
      // (k-step, tile) blocks in order; the next block's hi|lo fragments are read one block ahead and
      // the schedule is pinned per block so that the compiler does not hoist a whole chunk of LDS reads
      f32x4 xq[KSC][2];
#pragma unroll
      for (int k = 0; k < KSC; ++k) { xq[k][0] = xl[(k * 2 + 0) * 64]; xq[k][1] = xl[(k * 2 + 1) * 64]; }
      h8 fr[2][2];
      fr[0][0] = wl[0];
      fr[0][1] = wl[64];
      wn_static_for<KSC * JT>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;
        constexpr int k = blk / JT, j = blk % JT;
        if constexpr (blk + 1 < KSC * JT) {
          fr[(blk + 1) & 1][0] = wl[((blk + 1) * 2 + 0) * 64];
          fr[(blk + 1) & 1][1] = wl[((blk + 1) * 2 + 1) * 64];
        }
        h8 bh, bl;
        wn_split8g(xq[k][0], xq[k][1], okv[k] ? sc : 0.f, bh, bl);       // masked rows / k-steps contribute zero
        acc[j] = wn_mfma16g(fr[blk & 1][1], bh, acc[j]);
        acc[j] = wn_mfma16g(fr[blk & 1][0], bl, acc[j]);
        acc[j] = wn_mfma16g(fr[blk & 1][0], bh, acc[j]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // this chunk's activation buffer is free again once its LDS reads have returned
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int c = 0; c < nchunks; c += 2) {
      if (c + 1 < nchunks) xdma(c + 1, okb);
      if (c + 2 < nchunks) wdma(c + 2);
      arrive(c);
      compute(c, oka);
      if (c + 1 >= nchunks) break;
      if (c + 2 < nchunks) xdma(c + 2, oka);
      if (c + 3 < nchunks) wdma(c + 3);
      arrive(c + 1);
      compute(c + 1, okb);
    }

    if (live) wn_g16_epilogue<JT, PITCH, 0, EPI>(a, acc, inv_sc, jb, b, t, row0, rows_valid, stage, lane, wmax);
  }
  if (absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) { if (a.absmax_any) wn_absmax_publish_any(absmax_out, wmax); else wn_absmax_publish(absmax_out, wmax); }
  }
}

// Wide form of the streamed kernel: 256 output channels per workgroup (JT = 8 row tiles per wave), so a
// contraction with N = 256 (the folded skip sum over all blocks' gated activations, K = N_blocks * D)
// reads its activations ONCE instead of once per 128-column block.  One k-step per chunk (8 blocks of
// 2 KiB = 16 KiB of weight fragments, ring of 4, two chunks in flight) and a ring of FOUR activation
// buffers per wave inside its output stage (three k-steps = 6 KiB per wave in flight: the operand stream
// comes from HBM, the weights from L2).  Every chunk issues exactly one weight request group and one
// activation request group -- past the end of K they are clamped re-reads into buffers nobody uses any
// more -- so the counted wait is the same constant everywhere:
//   issue order  w0 x0 w1 x1 x2 | w2 x3 | w3 x4 | ...   ->  "x(c), w(c) landed" = at most 2 PT + 3 PX newer.
// Per output element the MFMA sequence is the streamed kernel's (k ascending; lo*hi, hi*lo, hi*hi).
// Measured on the folded skip sum of configs[1] (K = 1920): 608 -> 430 us.  Knocking parts out of the loop
// shows its phases do not overlap yet (loop + epilogue without DMA / MFMA / weight reads 123 us, + weight
// fragment LDS reads 213, + MFMA 303, + DMA waits 430): the eight waves run in lock step behind the
// per-chunk barrier.  Short contractions (K < 512) stay on the 128-column kernel, whose two column blocks
// overlap each other's epilogue.
struct WnG16W {
  static constexpr int JT = 8, NBUF = 4, XB = 4, WAVES = 8;
  static constexpr int CHUNK_BYTES = JT * 2048;        // one k-step of 8 row tiles
  static constexpr int XBUF_BYTES = 2048;              // one k-step of a wave's 32 rows
  static constexpr int PITCH = 68;
  static constexpr int STAGE_BYTES = 32 * PITCH * 4;   // 8704 >= XB * XBUF_BYTES
  static constexpr int LDS_BYTES = NBUF * CHUNK_BYTES + WAVES * STAGE_BYTES;
};

template <int EPI>
__global__ __launch_bounds__(512, 1) void wn_gemm_rows16_wide_kernel(WnGemmArgs a, const float* w16, int nks_total,
                                                                     const float* absmax_in0, const float* absmax_in1,
                                                                     float* absmax_out) {
  using G = WnG16W;
  constexpr int JT = G::JT, PITCH = G::PITCH;
  static_assert(G::XB * G::XBUF_BYTES <= G::STAGE_BYTES, "activation ring must fit the output stage");
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  float* stage = reinterpret_cast<float*>(smem + G::NBUF * G::CHUNK_BYTES + wave * G::STAGE_BYTES);
  unsigned char* const xbuf = reinterpret_cast<unsigned char*>(stage);

  float sc = 1.0f, inv_sc = 1.0f;
  if (absmax_in0) {
    float m = *absmax_in0;
    if (absmax_in1) m = fmaxf(m, *absmax_in1);
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);
      e = max(-100, min(100, e));
      sc = ldexpf(1.0f, -e);
      inv_sc = ldexpf(1.0f, e);
    }
  }

  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  const int ny = (a.JTtot + JT - 1) / JT;
  const int gxw = gridDim.x / ny;
  const int cb = blockIdx.x % ny, xb = blockIdx.x / ny;
  const int64_t passes = (ntiles + (int64_t)gxw * G::WAVES - 1) / ((int64_t)gxw * G::WAVES);
  const int jb = cb * JT;

  static_assert(WN_MAXSEG == 4, "segment select below is written out for four segments");
  const int e0 = (a.seg[0].K + 15) >> 4;
  const int e1 = a.nseg > 1 ? e0 + ((a.seg[1].K + 15) >> 4) : nks_total;
  const int e2 = a.nseg > 2 ? e1 + ((a.seg[2].K + 15) >> 4) : nks_total;
  const int plane_ks0 = a.seg[0].plane_k > 0 ? a.seg[0].plane_k / 16 : 0;
  const int64_t plane_st0 = a.seg[0].plane_stride;
  float wmax = 0.f;
  constexpr int PT = G::CHUNK_BYTES / 16 / 512;        // 2 weight-DMA instructions per thread and chunk
  constexpr int PX = 2;                                // activation-DMA instructions per lane and chunk

  for (int64_t pass = 0; pass < passes; ++pass) {
    const int64_t tile = (pass * gxw + xb) * G::WAVES + wave;
    const bool live = tile < ntiles;
    const int b = live ? (int)(tile / tiles_per_b) : 0;
    const int t0 = live ? (int)(tile % tiles_per_b) * 32 : 0;
    const int t = t0 + tl;
    const int rows_valid = live ? min(32, a.T - t0) : 0;
    const int64_t row0 = (int64_t)b * a.T + t0;

    // activations of k-step c -> activation buffer c % 4; returns whether this lane's row takes part
    auto xdma = [&](int c) -> bool {
      const int ks = min(c, nks_total - 1);
      const int si = (ks >= e0) + (ks >= e1) + (ks >= e2);              // wave-uniform segment index
      const int kk = ks - (si == 0 ? 0 : si == 1 ? e0 : si == 2 ? e1 : e2);
      const float* bx = a.seg[si].x;
      const int ld = a.seg[si].ldx, sh = a.seg[si].shift;
      const int ts = t - sh;
      const bool ok = live && t < a.T && ts >= 0 && ts < a.T;
      const float* src = bx + ((int64_t)b * a.T + (ok ? ts : 0)) * ld + 4 * h + 16 * kk;
      if (plane_ks0 > 0 && si == 0) src += (int64_t)(kk / plane_ks0) * plane_st0 - (int64_t)(kk / plane_ks0) * plane_ks0 * 16;
      unsigned char* dst = xbuf + (c & (G::XB - 1)) * G::XBUF_BYTES;    // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8),
                                       (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
      return ok && c < nks_total;
    };
    auto wdma = [&](int c) {
      const int ks = min(c, nks_total - 1);
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const int f = tid + 512 * i;
        const int blk = f >> 7, within = f & 127;
        const int j = min(jb + blk, a.JTtot - 1);
        const f32x4* src = reinterpret_cast<const f32x4*>(w16) + ((int64_t)ks * a.JTtot + j) * 128 + within;
        unsigned char* dst = smem + (c & (G::NBUF - 1)) * G::CHUNK_BYTES + (512 * i + wave * 64) * 16;   // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };

    f32x16 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    bool okr[G::XB];
    __syncthreads();                                   // previous pass finished with the ring and this stage
    wdma(0);
    okr[0] = xdma(0);
    wdma(1);
    okr[1] = xdma(1);
    okr[2] = xdma(2);
    for (int c0 = 0; c0 < nks_total; c0 += G::XB) {
      wn_static_for<G::XB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int c = c0 + i;
        if (c < nks_total) {                           // workgroup-uniform
          wdma(c + 2);
          okr[(i + 3) & 3] = xdma(c + 3);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT + 3 * PX) : "memory");
          asm volatile("s_barrier" ::: "memory");
          const h8* wl = reinterpret_cast<const h8*>(smem + (i & (G::NBUF - 1)) * G::CHUNK_BYTES) + lane;
          const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + i * G::XBUF_BYTES) + lane;
          const f32x4 x0 = xl[0], x1 = xl[64];
          h8 fr[2][2];
          fr[0][0] = wl[0];
          fr[0][1] = wl[64];
          h8 bh, bl;
          wn_split8g(x0, x1, okr[i] ? sc : 0.f, bh, bl);               // masked rows / k-steps contribute zero
          wn_static_for<JT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < JT) {
              fr[(j + 1) & 1][0] = wl[((j + 1) * 2 + 0) * 64];
              fr[(j + 1) & 1][1] = wl[((j + 1) * 2 + 1) * 64];
            }
            acc[j] = wn_mfma16g(fr[j & 1][1], bh, acc[j]);
            acc[j] = wn_mfma16g(fr[j & 1][0], bl, acc[j]);
            acc[j] = wn_mfma16g(fr[j & 1][0], bh, acc[j]);
            __builtin_amdgcn_sched_barrier(0);
          });
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this chunk's buffers are free again
        }
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail requests land before the stage is reused
    if (live) wn_g16_epilogue<JT, PITCH, 0, EPI>(a, acc, inv_sc, jb, b, t, row0, rows_valid, stage, lane, wmax);
  }
  if (absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) { if (a.absmax_any) wn_absmax_publish_any(absmax_out, wmax); else wn_absmax_publish(absmax_out, wmax); }
  }
}

// Thin form for the few-row contractions of queued generation (rows = utterances, <= a few hundred):
// the streamed kernel would run them in ONE chunk-synchronised workgroup.  Here every (32-row tile,
// 32-column tile) pair is its own single-wave workgroup that walks the whole K range with the weight
// fragments and the activations of the next PD k-steps in flight in a register ring.  The per-element
// MFMA sequence (k order, lo*hi, hi*lo, hi*hi) is exactly the streamed kernel's, so the results are
// bit-identical to what the sliding-window path computes for the same rows.
template <int PD>
__global__ __launch_bounds__(64) void wn_gemm_rows16_thin_kernel(WnGemmArgs a, const float* w16, int nks_total,
                                                                 const float* absmax_in0, const float* absmax_in1,
                                                                 float* absmax_out) {
  constexpr int PITCH = 68;
  __shared__ __attribute__((aligned(16))) float stage[32 * PITCH];
  const int lane = threadIdx.x & 63;
  const int tl = lane & 31, h = lane >> 5;
  float sc = 1.0f, inv_sc = 1.0f;
  if (absmax_in0) {
    float m = *absmax_in0;
    if (absmax_in1) m = fmaxf(m, *absmax_in1);
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);
      e = max(-100, min(100, e));
      sc = ldexpf(1.0f, -e);
      inv_sc = ldexpf(1.0f, e);
    }
  }
  const int tiles_per_b = (a.T + 31) >> 5;
  const int tile = blockIdx.x;
  const int jb = blockIdx.y;
  const int b = tile / tiles_per_b;
  const int t0 = (tile % tiles_per_b) * 32;
  const int t = t0 + tl;
  const int rows_valid = min(32, a.T - t0);
  const int64_t row0 = (int64_t)b * a.T + t0;

  int ks_end[WN_MAXSEG];
  {
    int acc = 0;
#pragma unroll
    for (int s = 0; s < WN_MAXSEG; ++s) {
      if (s < a.nseg) acc += (a.seg[s].K + 15) >> 4;
      ks_end[s] = acc;
    }
  }
  const float* xrow_s[WN_MAXSEG];
  bool xok_s[WN_MAXSEG];
  const int plane_ks0 = a.seg[0].plane_k > 0 ? a.seg[0].plane_k / 16 : 0;
  const int64_t plane_st0 = a.seg[0].plane_stride;
#pragma unroll
  for (int s = 0; s < WN_MAXSEG; ++s) {
    xrow_s[s] = nullptr;
    xok_s[s] = false;
    if (s < a.nseg) {
      const int ts = t - a.seg[s].shift;
      xok_s[s] = t < a.T && ts >= 0 && ts < a.T;
      xrow_s[s] = a.seg[s].x + ((int64_t)b * a.T + (xok_s[s] ? ts : 0)) * a.seg[s].ldx + 4 * h;
    }
  }
  auto load_x = [&](int ks, f32x4& q0, f32x4& q1) {
    q0 = f32x4{0.f, 0.f, 0.f, 0.f};
    q1 = q0;
    if (ks >= nks_total) return;
    const float* xr = xrow_s[0];
    bool ok = xok_s[0];
    int kk = ks;
#pragma unroll
    for (int s = 1; s < WN_MAXSEG; ++s)
      if (ks >= ks_end[s - 1]) { xr = xrow_s[s]; ok = xok_s[s]; kk = ks - ks_end[s - 1]; }
    if (ok) {
      if (plane_ks0 > 0 && ks < ks_end[0]) xr += (int64_t)(kk / plane_ks0) * plane_st0 - (int64_t)(kk / plane_ks0) * plane_ks0 * 16;
      q0 = wn_ldg4(xr + 16 * kk);
      q1 = wn_ldg4(xr + 16 * kk + 8);
    }
  };
  // hi|lo fragments of (k-step, column tile jb); k-steps past the end are clamped (their activations are zero)
  const h8* wbase = reinterpret_cast<const h8*>(w16) + lane;
  auto load_w = [&](int ks, h8& hi, h8& lo) {
    const int64_t blk = (int64_t)min(ks, nks_total - 1) * a.JTtot + jb;
    hi = wbase[(blk * 2 + 0) * 64];
    lo = wbase[(blk * 2 + 1) * 64];
  };

  f32x16 acc[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
  h8 wf[PD][2];
  f32x4 xv[PD][2];
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    load_w(i, wf[i][0], wf[i][1]);
    load_x(i, xv[i][0], xv[i][1]);
  }
  for (int ks0 = 0; ks0 < nks_total; ks0 += PD) {
    wn_static_for<PD>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      h8 bh, bl;
      wn_split8g(xv[i][0], xv[i][1], sc, bh, bl);
      acc[0] = wn_mfma16g(wf[i][1], bh, acc[0]);
      acc[0] = wn_mfma16g(wf[i][0], bl, acc[0]);
      acc[0] = wn_mfma16g(wf[i][0], bh, acc[0]);
      load_w(ks0 + i + PD, wf[i][0], wf[i][1]);
      load_x(ks0 + i + PD, xv[i][0], xv[i][1]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  float wmax = 0.f;
  wn_g16_epilogue<1, PITCH>(a, acc, inv_sc, jb, b, t, row0, rows_valid, stage, lane, wmax);
  if (absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) { if (a.absmax_any) wn_absmax_publish_any(absmax_out, wmax); else wn_absmax_publish(absmax_out, wmax); }
  }
}

// Resident form for contractions whose whole hi|lo weight image fits LDS next to the stages
// (<= 80 KiB: the per-block backward products d z and d x): weights are loaded once per
// persistent workgroup and the tile loop has NO barrier, so the 8 waves drift apart and one wave's
// epilogue / load latency hides under the others' MFMAs (same structure as wn_layer16.hip).
template <int JT>
struct WnG16R {
  static constexpr int MAX_W_BYTES = 81920;
  static constexpr int PITCH = 68;
  static constexpr int STAGE_BYTES = 32 * PITCH * 4;
  static constexpr int WAVES = 8;
  static constexpr int LDS_BYTES = MAX_W_BYTES + WAVES * STAGE_BYTES;       // 151552
  static constexpr int PF = 4;                                              // k-steps of activations in flight
};

// PREK: which epilogue row operands are fetched ahead: 0 none, 1 addc, 2 gate (tanh | sigmoid)
template <int JT, int PREK>
__global__ __launch_bounds__(512, 2) void wn_gemm_rows16_resident_kernel(WnGemmArgs a, const float* w16, int nks_total,
                                                                         const float* absmax_in0, const float* absmax_in1,
                                                                         float* absmax_out) {
  using G = WnG16R<JT>;
  constexpr int PITCH = G::PITCH, PF = G::PF;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  float* stage = reinterpret_cast<float*>(smem + G::MAX_W_BYTES + wave * G::STAGE_BYTES);
  {
    const int n16 = nks_total * JT * 128;                // 16-byte pieces of the image (JTtot == JT)
    // (every load in flight before the first LDS store: wn_images_to_lds)
    wn_images_to_lds<512, G::MAX_W_BYTES / 16>(w16, smem, n16, nullptr, nullptr, 0, tid);
  }
  __syncthreads();
  const h8* wl = reinterpret_cast<const h8*>(smem) + lane;

  float sc = 1.0f, inv_sc = 1.0f;
  if (absmax_in0) {
    float m = *absmax_in0;
    if (absmax_in1) m = fmaxf(m, *absmax_in1);
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);
      e = max(-100, min(100, e));
      sc = ldexpf(1.0f, -e);
      inv_sc = ldexpf(1.0f, e);
    }
  }
  // Segment fields are read straight from the kernel-argument block by wave-uniform selects (scalar
  // loads).  Copies of them in local arrays or captured locals end up in scratch memory once the
  // k-step that indexes them is a runtime value, and every activation load then waits for a scratch
  // load first (s_waitcnt vmcnt(0) per load).
  static_assert(WN_MAXSEG == 4, "segment select below is written out for four segments");
  const int e0 = (a.seg[0].K + 15) >> 4;
  const int e1 = a.nseg > 1 ? e0 + ((a.seg[1].K + 15) >> 4) : nks_total;
  const int e2 = a.nseg > 2 ? e1 + ((a.seg[2].K + 15) >> 4) : nks_total;
  float wmax = 0.f;
  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  struct TileCtx {
    int b, t, rows_valid;
    int64_t row0;
  };
  auto make_ctx = [&](int64_t tile, TileCtx& c) {
    c.b = (int)(tile / tiles_per_b);
    const int t0 = (int)(tile % tiles_per_b) * 32;
    c.t = t0 + tl;
    c.rows_valid = min(32, a.T - t0);
    c.row0 = (int64_t)c.b * a.T + t0;
  };
  // straight-line, unconditional (k-step inside [0, nks_total): the launcher guarantees nks_total % PF == 0;
  // masked rows read a clamped row and are zeroed where they are consumed)
  auto load_x = [&](const TileCtx& c, int ks, f32x4& q0, f32x4& q1, bool& okout) {
    const int si = (ks >= e0) + (ks >= e1) + (ks >= e2);               // wave-uniform segment index
    const int kk = ks - (si == 0 ? 0 : si == 1 ? e0 : si == 2 ? e1 : e2);
    const float* bx = a.seg[si].x;
    const int ld = a.seg[si].ldx, sh = a.seg[si].shift;
    const int ts = c.t - sh;
    const bool ok = c.t < a.T && ts >= 0 && ts < a.T;
    const float* src = bx + ((int64_t)c.b * a.T + (ok ? ts : 0)) * ld + 4 * h + 16 * kk;
    q0 = wn_ldg4(src);
    q1 = wn_ldg4(src + 8);
    okout = ok;
  };
  const int64_t tstride = (int64_t)gridDim.x * G::WAVES;
  int64_t tile = (int64_t)blockIdx.x * G::WAVES + wave;
  // Rolling activation ring: slot k holds k-step (i mod PF == k).  Per k-step: the slot is turned into
  // fp16 hi | lo operands (the compiler's counted s_waitcnt leaves the PF - 1 newer requests in flight),
  // re-issued for k-step i + PF -- of this tile or, past its end, of the wave's next tile -- and only
  // then the MFMAs run.  PF - 1 k-steps are in flight at all times, tile boundaries included.  All loads
  // are straight-line and unconditional: a branch around a load makes hipcc wait vmcnt(0) per load.
  TileCtx cur, nxt;
  f32x4 xr_[PF][2];
  bool okr[PF];
  if (tile < ntiles) {
    make_ctx(tile, cur);
#pragma unroll
    for (int k = 0; k < PF; ++k) load_x(cur, k, xr_[k][0], xr_[k][1], okr[k]);
  }
  for (; tile < ntiles; tile += tstride) {
    const bool has_next = tile + tstride < ntiles;     // wave-uniform
    if (has_next) make_ctx(tile + tstride, nxt);
    // ---- epilogue operands of THIS tile, issued before the contraction (latency hidden under it) ----
    WnG16Pre<JT> pre;
    {
      // rows past the end of the utterance read the tile's first row: the epilogue never uses them
      const int64_t row = cur.row0 + (cur.t < a.T ? tl : 0);
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int n0 = 32 * j + 8 * rq + 4 * h;
          if constexpr (PREK == 1) pre.addc[j][rq] = wn_ldg4(a.addc + row * a.ld_addc + n0);
          if constexpr (PREK == 2) {
            pre.aux[j][rq] = wn_ldg4(a.aux + row * a.ld_aux + n0);
            pre.aux2[j][rq] = wn_ldg4(a.aux2 + row * a.ld_aux2 + n0);
          }
        }
    }
    f32x16 acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int ks0 = 0; ks0 < nks_total; ks0 += PF) {
      wn_static_for<PF>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        h8 bh, bl;
        wn_split8g(xr_[k][0], xr_[k][1], okr[k] ? sc : 0.f, bh, bl);    // masked rows contribute zero
        __builtin_amdgcn_sched_barrier(0);
        const int ksn = ks0 + k + PF;
        if (ksn < nks_total) load_x(cur, ksn, xr_[k][0], xr_[k][1], okr[k]);
        else if (has_next) load_x(nxt, ksn - nks_total, xr_[k][0], xr_[k][1], okr[k]);
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const h8 ah = wl[(((ks0 + k) * JT + j) * 2 + 0) * 64];
          const h8 al = wl[(((ks0 + k) * JT + j) * 2 + 1) * 64];
          acc[j] = wn_mfma16g(al, bh, acc[j]);
          acc[j] = wn_mfma16g(ah, bl, acc[j]);
          acc[j] = wn_mfma16g(ah, bh, acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    constexpr int EPI = PREK == 2 ? WN_EPI_GATE_BWD : (PREK == 1 ? WN_EPI_PLAIN : -1);
    wn_g16_epilogue<JT, PITCH, PREK, EPI>(a, acc, inv_sc, 0, cur.b, cur.t, cur.row0, cur.rows_valid, stage, lane, wmax, &pre);
    if (has_next) cur = nxt;
  }
  if (absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) { if (a.absmax_any) wn_absmax_publish_any(absmax_out, wmax); else wn_absmax_publish(absmax_out, wmax); }
  }
}

// eligibility of a rows GEMM for the split-precision kernel (otherwise the fp32 kernel runs)
int wn_gemm_rows16_ok(const WnGemmArgs& a) {
  // (32 output channels: the image must be padded to two row tiles -- JTtot = 2, the second tile all zero -- and the
  // kernels compute 64 columns of which the epilogue stores 32: the products are not what bounds these launches)
  if (a.N % 32 != 0 || (a.N < 64 && a.JTtot < 2)) return 0;
  if (!a.vec_out) return 0;
  for (int s = 0; s < a.nseg; ++s) {
    if (a.seg[s].K % 16 != 0 || !a.seg[s].vec) return 0;
    if (a.seg[s].plane_k > 0 && (s != 0 || a.seg[s].plane_k % 16 != 0)) return 0;
  }
  if (a.rowbias && (a.ld_rowbias % 4 != 0)) return 0;
  return 1;
}

int wn_launch_gemm_rows16(const WnGemmArgs& a, const float* w16, const float* absmax_in0,
                          const float* absmax_in1, float* absmax_out, hipStream_t s) {
  if (a.B <= 0 || a.T <= 0 || a.N <= 0) return WN_OK;
  int nks = 0;
  for (int i = 0; i < a.nseg; ++i) nks += (a.seg[i].K + 15) / 16;
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  int64_t gx = (tiles + 7) / 8;
  if (gx > 256) gx = 256;
  const int jt_need = (a.N + 31) / 32;
  if (a.epi == WN_EPI_GATE_FWD && (jt_need % 4 != 0 || !a.bias)) { wn_set_error("gemm_rows16: gate-forward epilogue needs 128-column blocks and a bias"); return WN_E_UNSUPPORTED; }
  // few rows (queued generation): one wave per (row tile, column tile), see the thin kernel
  if (tiles * jt_need <= 64 && tiles <= 8 && a.epi != WN_EPI_GATE_FWD) {
    hipLaunchKernelGGL(wn_gemm_rows16_thin_kernel<8>, dim3((unsigned)tiles, (unsigned)jt_need), dim3(64), 0, s, a, w16, nks,
                       absmax_in0, absmax_in1, absmax_out);
    WN_HIP_CHECK(hipGetLastError());
    return WN_OK;
  }
  // long contractions with 256 output channels over enough rows to fill the chip with one column block (fewer rows,
  // e.g. the sliding-window sampler's: two column blocks give twice the workgroups)
  if (jt_need == 8 && a.JTtot == 8 && a.epi == WN_EPI_PLAIN && nks >= 32 && tiles >= 2048) {
    hipLaunchKernelGGL((wn_gemm_rows16_wide_kernel<WN_EPI_PLAIN>), dim3((unsigned)gx), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    WN_HIP_CHECK(hipGetLastError());
    return WN_OK;
  }
  if (jt_need <= 2 && a.JTtot == 2 && (int64_t)nks * 2 * 2048 <= WnG16R<2>::MAX_W_BYTES &&
      nks % WnG16R<2>::PF == 0 &&
      a.seg[0].plane_k == 0) {
    // (the forms that fetch their epilogue operands ahead read all 64 columns of them: not for a padded 32-column product)
    if (jt_need == 1)
      hipLaunchKernelGGL((wn_gemm_rows16_resident_kernel<2, 0>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else if (a.epi == WN_EPI_GATE_BWD && !a.addc)
      hipLaunchKernelGGL((wn_gemm_rows16_resident_kernel<2, 2>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else if (a.epi == WN_EPI_PLAIN && a.addc)
      hipLaunchKernelGGL((wn_gemm_rows16_resident_kernel<2, 1>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else
      hipLaunchKernelGGL((wn_gemm_rows16_resident_kernel<2, 0>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
  } else if (jt_need <= 2) {
    // the epilogue kind is a template parameter (no dead code of the other kinds in the kernel)
    if (a.epi == WN_EPI_PLAIN) hipLaunchKernelGGL((wn_gemm_rows16_kernel<2, WN_EPI_PLAIN>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else if (a.epi == WN_EPI_DACT) hipLaunchKernelGGL((wn_gemm_rows16_kernel<2, WN_EPI_DACT>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else hipLaunchKernelGGL((wn_gemm_rows16_kernel<2, WN_EPI_GATE_BWD>), dim3((unsigned)gx, 1), dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
  } else {
    // 128 output channels per workgroup column; wider outputs re-read the activations per column
    const dim3 g4((unsigned)(gx * ((a.JTtot + 3) / 4)), 1);
    if (a.epi == WN_EPI_GATE_FWD) hipLaunchKernelGGL((wn_gemm_rows16_kernel<4, WN_EPI_GATE_FWD>), g4, dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else if (a.epi == WN_EPI_PLAIN) hipLaunchKernelGGL((wn_gemm_rows16_kernel<4, WN_EPI_PLAIN>), g4, dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else if (a.epi == WN_EPI_DACT) hipLaunchKernelGGL((wn_gemm_rows16_kernel<4, WN_EPI_DACT>), g4, dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
    else hipLaunchKernelGGL((wn_gemm_rows16_kernel<4, WN_EPI_GATE_BWD>), g4, dim3(512), 0, s, a, w16, nks, absmax_in0, absmax_in1, absmax_out);
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
