// Weight gradients with the transposition in the LDS READ (gfx950, split-precision MFMA).  First use: a 128-channel
// block's gated conv, BOTH taps from one read of du:
//   dW_d[tap][k][n] = sum_t x[t - (1 - tap) d][k] * du[t][n],   db_d[n] = sum_t du[t][n]       (src/layers.py:82-88 reversed)
// with k < 128, n < 256: a 256 x 256 output block (the two taps stacked: the flat layout is tap-major) per workgroup.
//
// The staged kernels (wn_wgrad_layer.hip, wn_wgrad_pair.hip) transpose at LOAD time: a thread fetches 8 consecutive time
// steps of one channel with 8 dword loads, so that time becomes the MFMA K dimension.  That costs 8 x the load
// instructions of a row-wise read and two 8-register buffers per operand stream; with the 128 accumulator registers of a
// both-taps tile it spills (wn_wgrad_pair_kernel<.., DUAL>: 80 VGPRs, 22.1 vs 16.4 ms per step).  Here every operand is
// read ROW-wise with 16-byte loads (a wave instruction = one or two whole rows), split once into fp16 hi | lo planes kept
// row-major [time][channel] in LDS, and the transposition happens in the LDS read: ds_read_b64_tr_b16 hands a lane 4
// consecutive TIME steps of its channel (lane map checked by tools/tr_probe.hip).  A row of a plane is padded by 64 bytes,
// so the four rows one transposed read touches fall on four different bank quarters.
//
// One workgroup of 8 waves per (block, utterance, time range); chunks of 32 time steps (two MFMA k-steps), two LDS
// stages, one barrier per chunk: the rows of the next TWO chunks are in registers (two register sets, each stored to its
// stage and re-requested at once) while this chunk's 48 products per wave run -- a request then has two product phases
// (~1.5 us) to land instead of one: with one chunk in flight the kernel waited on every chunk (3.5-4.0 TB/s).
// Wave w owns output row tiles {2 (w & 3), +1} x column tiles {4 (w >> 2) .. +3}: 8 accumulator tiles.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 wt_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 wt_h4 __attribute__((ext_vector_type(4)));
typedef short wt_s4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WT_ROWS = 32;                                            // time steps per chunk

__device__ __forceinline__ f32x4 wt_ldg4(const float* p) { return *(const __attribute__((address_space(1))) f32x4*)(p); }

// 8 consecutive time steps (rows 16 ks + 8 h .. + 7) of channel chbase + (lane & 31): two transposed reads of 4
template <int PITCH>
__device__ __forceinline__ wt_h8 wt_frag(const unsigned char* lanebase, int off) {
  const wt_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s4*)(lanebase + off));
  const wt_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wt_s4*)(lanebase + off + 4 * PITCH));
  wt_h8 r;
  const wt_h4 ha = *reinterpret_cast<const wt_h4*>(&a), hb = *reinterpret_cast<const wt_h4*>(&b);
  r[0] = ha[0]; r[1] = ha[1]; r[2] = ha[2]; r[3] = ha[3]; r[4] = hb[0]; r[5] = hb[1]; r[6] = hb[2]; r[7] = hb[3];
  return r;
}

__device__ __forceinline__ void wt_split4(const f32x4& v, float s, wt_h4& hi, wt_h4& lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 hh = (_Float16)(f[e] * s);
    hi[e] = hh;
    lo[e] = (_Float16)__builtin_fmaf(f[e], s, -(float)hh);
  }
}

}  // namespace

// KC = channels of X (per tap), NC = channels of G, TAPS = 2: X is x[t - shift] | x[t] (both taps of a kernel-size-2
// conv, stacked along the output's k index); TAPS = 1: one product with row shift `shift`.  LDX / LDG / LDW: row strides of
// X, G and of the dW matrix (G and dW may be a column half of a wider tensor).  TKW x TNW = output tiles per wave.
// G2: G is two tensors of NC / 2 channels each (J.g_off | J.g2_off, row stride LDG both); the lower half's product goes to
// (slab, w_off, b_off) as usual, the upper half's to (slab2, w2_off, b2_off), both with row pitch LDW.
// XSEG > 0 (TAPS = 1, not G2): the KC channels of X are KC / XSEG tensors of XSEG channels each (row stride LDX = XSEG),
// J.g2_off floats apart, of which the first J.pad_ exist (the rest reads as zero and its output rows are not written):
// the gated activations z of several blocks (block-major Z) against ONE read of G -- the folded skip path's
// M = Z^T dL/da (src/layers.py:216-217 and model.py:105-111 reversed) for 64- and 32-channel blocks.
template <int KC, int NC, int TAPS, int LDX, int LDG, int LDW, int TKW, int TNW, bool G2 = false, int XSEG = 0>
__global__ __launch_bounds__(512, 2) void wn_wgrad_tr_kernel(const WnWgPair* jobs, float* ws, float* slab, int64_t P, int B,
                                                             int T, int spb, float* slab2, int64_t P2) {
  constexpr int XC = TAPS * KC, NCH = XC + NC;                  // LDS channels: x taps | g
  constexpr int PITCH = NCH * 2 + 64;                           // bytes per time row of a plane (the pad spreads 4 rows over the banks)
  constexpr int PLANE = WT_ROWS * PITCH, STAGE = 2 * PLANE;
  constexpr int KT = XC / 32, NT = NC / 32;
  static_assert((KT / TKW) * (NT / TNW) == 8 && KT % TKW == 0 && NT % TNW == 0, "8 waves tile the output block");
  static_assert((PITCH / 4) % 64 == 16, "row pitch must rotate the banks by a quarter");
  static_assert(XSEG == 0 || (TAPS == 1 && !G2 && LDX == XSEG && KC % XSEG == 0 && XSEG % 32 == 0), "segmented X");
  constexpr int XPR = KC / 4, GPR = NC / 4;                     // 16-byte pieces per row
  constexpr int XRP = 512 / XPR, GRP = 512 / GPR;               // rows per pass of the 512 threads
  constexpr int XP = WT_ROWS / XRP, GP = WT_ROWS / GRP;         // passes per chunk
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  static_assert(2 * STAGE <= 160 * 1024 && GRP * NC * 4 <= 2 * STAGE, "LDS");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  const WnWgPair J = jobs[blockIdx.y];
  const int split = blockIdx.x;
  const int ub = split / spb, sp = split % spb;
  int len = (T + spb - 1) / spb;
  len = (len + WT_ROWS - 1) & ~(WT_ROWS - 1);
  const int r0 = sp * len, r1 = min(T, r0 + len);
  const int d = J.shift;

  auto pow2 = [&](int64_t off, float& sc, float& iv) {
    sc = 1.0f; iv = 1.0f;
    if (off >= 0) {
      const float m = ws[off];
      if (m > 0.f && m < 3.0e38f) {
        int e;
        (void)frexpf(m, &e);
        e = max(-100, min(100, e));
        sc = ldexpf(1.0f, -e);
        iv = ldexpf(1.0f, e);
      }
    }
  };
  float gsc, inv, gsc2 = 1.0f, inv2 = 1.0f;
  pow2(J.gmax_off, gsc, inv);
  if constexpr (G2) pow2(J.gmax2_off, gsc2, inv2);

  // ---- this thread's pieces of a chunk: rows xr + XRP k of every x tap (4 channels at xc), rows gr + GRP k of g (4 at gc) ----
  const int xr = tid / XPR, xc = (tid % XPR) * 4;
  const int gr = tid / GPR, gc = (tid % GPR) * 4;
  const int nseg = XSEG > 0 ? J.pad_ : 0;
  const bool xvalid = XSEG == 0 || xc / (XSEG > 0 ? XSEG : 1) < nseg;
  const float* xbase = XSEG > 0 ? ws + J.x_off + (int64_t)(xvalid ? xc / (XSEG > 0 ? XSEG : 1) : 0) * J.g2_off + (int64_t)ub * T * LDX + xc % (XSEG > 0 ? XSEG : 1)
                                : ws + J.x_off + (int64_t)ub * T * LDX + xc;
  const bool upper = G2 && gc >= NC / 2;                 // this thread's g piece belongs to the second tensor
  const float* gbase = upper ? ws + J.g2_off + (int64_t)ub * T * LDG + (gc - NC / 2) : ws + J.g_off + (int64_t)ub * T * LDG + gc;
  const float gs_t = upper ? gsc2 : gsc;
  struct Regs { f32x4 xs[TAPS][XP]; f32x4 gv[GP]; };     // one chunk in flight: x[t - shift] (| x[t]), g
  Regs ra, rb;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // Every request is UNCONDITIONAL (row index clamped into the utterance, rows outside the range zeroed when the chunk is
  // stored): the number of loads per iteration is then a constant and hipcc can wait for the OLDER register set with a
  // counted vmcnt(N) while the younger one stays in flight (behind a conditional load it falls back to vmcnt(0)).
  auto load = [&](int t0, Regs& q) {
    f32x4 (&xs)[TAPS][XP] = q.xs;
    f32x4 (&gv)[GP] = q.gv;
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int t = t0 + xr + XRP * k;
      const int tc = min(t, T - 1);
      if (XSEG > 0) {
        xs[0][k] = wt_ldg4(xbase + (int64_t)tc * LDX);
      } else {
        xs[0][k] = wt_ldg4(xbase + (int64_t)max(tc - d, 0) * LDX);
        if constexpr (TAPS == 2) xs[1][k] = wt_ldg4(xbase + (int64_t)tc * LDX);
      }
    }
#pragma unroll
    for (int k = 0; k < GP; ++k) {
      const int t = min(t0 + gr + GRP * k, T - 1);
      gv[k] = wt_ldg4(gbase + (int64_t)t * LDG);
    }
  };
  auto store = [&](int stage, const Regs& q, int t0) {
    const f32x4 (&xs)[TAPS][XP] = q.xs;
    const f32x4 (&gv)[GP] = q.gv;
    unsigned char* st = smem + stage * STAGE;
    const bool interior = (t0 + WT_ROWS <= r1) && (t0 - d >= 0);       // workgroup-uniform: no row masks
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      unsigned char* row = st + (xr + XRP * k) * PITCH;
      const int t = t0 + xr + XRP * k;
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        wt_h4 hi, lo;
        // (the first plane is the SHIFTED tap x[t - d] unless the rows are segments of an unshifted tensor)
        const bool ok = interior || (t < r1 && (XSEG > 0 || (TAPS == 2 && tp == TAPS - 1) || t - d >= 0));
        const f32x4 xv = (ok && xvalid) ? xs[tp][k] : zero4;
        wt_split4(xv, 1.0f, hi, lo);
        *reinterpret_cast<wt_h4*>(row + (tp * KC + xc) * 2) = hi;
        *reinterpret_cast<wt_h4*>(row + PLANE + (tp * KC + xc) * 2) = lo;
      }
    }
#pragma unroll
    for (int k = 0; k < GP; ++k) {
      wt_h4 hi, lo;
      unsigned char* row = st + (gr + GRP * k) * PITCH;
      const f32x4 g4 = (interior || t0 + gr + GRP * k < r1) ? gv[k] : zero4;
      wt_split4(g4, gs_t, hi, lo);
      bsum[0] += g4.x; bsum[1] += g4.y; bsum[2] += g4.z; bsum[3] += g4.w;
      *reinterpret_cast<wt_h4*>(row + (XC + gc) * 2) = hi;
      *reinterpret_cast<wt_h4*>(row + PLANE + (XC + gc) * 2) = lo;
    }
  };

  // ---- this wave's output tiles ----
  constexpr int WKD = KT / TKW;
  const int kt0 = TKW * (wave % WKD), nt0 = TNW * (wave / WKD);
  f32x16 acc[TKW][TNW];
#pragma unroll
  for (int i = 0; i < TKW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // lane part of every transposed read: row 8 h + q, columns 16 g + 4 p (q = (lane & 15) >> 2, p = lane & 3, g = (lane >> 4) & 1)
  const int lb = (8 * h + ((lane & 15) >> 2)) * PITCH + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  auto compute = [&](int stage) {
    const unsigned char* base = smem + stage * STAGE + lb;
    wn_static_for<2>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int ro = ks * 16 * PITCH;
      wt_h8 ah[TKW], al[TKW];
#pragma unroll
      for (int i = 0; i < TKW; ++i) {
        ah[i] = wt_frag<PITCH>(base, ro + 32 * (kt0 + i) * 2);
        al[i] = wt_frag<PITCH>(base, ro + PLANE + 32 * (kt0 + i) * 2);
      }
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const wt_h8 bh = wt_frag<PITCH>(base, ro + (XC + 32 * (nt0 + j)) * 2);
        const wt_h8 bl = wt_frag<PITCH>(base, ro + PLANE + (XC + 32 * (nt0 + j)) * 2);
#pragma unroll
        for (int i = 0; i < TKW; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
      }
    });
  };

  // ---- pipeline: this chunk in LDS, one barrier per chunk; the rows of the next two chunks in registers (ra, rb), or of
  //      the next one only where two register sets do not fit beside the accumulators (both taps: 343 spilled VGPRs) ----
  constexpr bool DEEP = TAPS == 1;
  if constexpr (DEEP) {
    load(r0, ra);
    load(r0 + WT_ROWS, rb);
    store(0, ra, r0);
    load(r0 + 2 * WT_ROWS, ra);
    __syncthreads();
    // invariant at the top of an iteration for chunk t0 (stage 0): rb holds chunk t0 + 32, ra holds chunk t0 + 64
    for (int t0 = r0; t0 < r1; t0 += 2 * WT_ROWS) {
      store(1, rb, t0 + WT_ROWS);                          // (its stage was read one barrier ago)
      load(t0 + 3 * WT_ROWS, rb);
      compute(0);
      __syncthreads();
      if (t0 + WT_ROWS >= r1) break;
      // chunk t0 + 32 (stage 1): ra holds chunk t0 + 64, rb chunk t0 + 96
      store(0, ra, t0 + 2 * WT_ROWS);
      load(t0 + 4 * WT_ROWS, ra);
      compute(1);
      __syncthreads();
    }
  } else {
    load(r0, ra);
    store(0, ra, r0);
    load(r0 + WT_ROWS, ra);
    __syncthreads();
    int st = 0;
    for (int t0 = r0; t0 < r1; t0 += WT_ROWS, st ^= 1) {
      store(st ^ 1, ra, t0 + WT_ROWS);                     // (registers hold chunk t0 + 32; its stage was read one barrier ago)
      load(t0 + 2 * WT_ROWS, ra);
      compute(st);
      __syncthreads();
    }
  }

  // ---- partial results -> this split's slab row (laid out like the flat gradient buffer; taps stacked along k) ----
  float* row = slab + (int64_t)split * P;
  float* row2 = G2 ? slab2 + (int64_t)split * P2 : nullptr;
#pragma unroll
  for (int i = 0; i < TKW; ++i) {
    if (XSEG > 0 && 32 * (kt0 + i) >= nseg * XSEG) continue;      // rows of a segment that does not exist (wave-uniform)
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int nt = nt0 + j;
      const bool up = G2 && nt >= NT / 2;                  // wave-uniform
      float* tbase = (up ? row2 + J.w2_off + 32 * (nt - NT / 2) : row + J.w_off + 32 * nt) + (int64_t)(32 * (kt0 + i)) * LDW + tl;
      const float iv = up ? inv2 : inv;
#pragma unroll
      for (int r = 0; r < 16; ++r) tbase[wn_drow(r, h) * LDW] = acc[i][j][r] * iv;
    }
  }
  if (J.b_off >= 0 || (G2 && J.b2_off >= 0)) {           // bias sums: the row groups hold partial sums of every column
    float* bpart = reinterpret_cast<float*>(smem);        // (every wave is past its last LDS read: the loop ends with a barrier)
#pragma unroll
    for (int e = 0; e < 4; ++e) bpart[gr * NC + gc + e] = bsum[e];
    __syncthreads();
    for (int n = tid; n < NC; n += 512) {
      float s8 = 0.f;
#pragma unroll
      for (int g = 0; g < GRP; ++g) s8 += bpart[g * NC + n];
      if (G2 && n >= NC / 2) { if (J.b2_off >= 0) row2[J.b2_off + n - NC / 2] = s8; }
      else if (J.b_off >= 0) row[J.b_off + n] = s8;
    }
  }
}

int wn_wgrad_tr_kind(int K, int N, int taps) {
  if (taps == 2) return (K == 128 && N == 256) ? 1 : 0;
  if (K == 128 && N == 128) return 2;
  if (K == 128 && N == 256) return 3;
  if (K == 256 && N == 128) return 4;
  if (K == 256 && N == 256) return 5;
  return 0;
}

int wn_launch_wgrad_tr(int kind, const WnWgPair* d_jobs, int njobs, float* ws, float* slab, int64_t P, int B, int T,
                       int splits_per_b, hipStream_t s, float* slab2, int64_t P2) {
  if (njobs <= 0) return WN_OK;
  const dim3 grid((unsigned)(B * splits_per_b), (unsigned)njobs);
#define WT_LAUNCH(...) hipLaunchKernelGGL((wn_wgrad_tr_kernel<__VA_ARGS__>), grid, dim3(512), 0, s, d_jobs, ws, slab, P, B, T, splits_per_b, slab2, P2)
  switch (kind) {
    case 1: WT_LAUNCH(128, 256, 2, 128, 256, 256, 2, 4); break;
    case 2: WT_LAUNCH(128, 128, 1, 128, 128, 128, 1, 2); break;
    case 3: WT_LAUNCH(128, 256, 1, 128, 256, 256, 1, 4); break;
    case 4: WT_LAUNCH(256, 128, 1, 256, 128, 128, 2, 2); break;
    case 5: WT_LAUNCH(256, 128, 1, 256, 256, 256, 2, 2); break;
    // dW_r = z^T g_o and M = z^T dL/da of a 128-channel block from one read of z (two 128-column products, pitch 128)
    case 6:
      if (!slab2) { wn_set_error("wgrad_tr: kind 6 needs the second slab"); return WN_E_INVALID; }
      WT_LAUNCH(128, 256, 1, 128, 128, 128, 1, 4, true);
      break;
    // M = Z^T dL/da: the z of four 64-channel (kind 7) or eight 32-channel blocks (kind 8) against one read of dL/da
    case 7: WT_LAUNCH(256, 128, 1, 64, 128, 128, 2, 2, false, 64); break;
    case 8: WT_LAUNCH(256, 128, 1, 32, 128, 128, 2, 2, false, 32); break;
    default: wn_set_error("wgrad_tr: unknown kind %d", kind); return WN_E_UNSUPPORTED;
  }
#undef WT_LAUNCH
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
