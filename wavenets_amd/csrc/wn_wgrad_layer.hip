// Weight gradients of one residual block per workgroup (gfx950): dW_d (both taps), db_d, dW_r, db_r
// of src/layers.py:66-97 from the tensors the backward pass already holds.
//   dW_d[tap][k][n] = sum_t x[t - (1 - tap) d][k] * du[t][n]      (KS = 2)
//   dW_r[k][n]      = sum_t z[t][k] * go[t][n]
//   db_d[n] = sum_t du[t][n],  db_r[n] = sum_t go[t][n]
// INNER instantiations (round 3): a non-gated conv of a stack deeper than 1 (src/layers.py:66-80, layers_per_block > 1):
// the same x[t-d] | x[t] staging against an output gradient of R channels, no 1x1 part.
// The generic batched kernel (wn_gemm.hip) runs one wave per (tap, K block, N block) job: every job
// re-reads and re-splits its operands (7.9 GB of loads per step at configs[1]) and waits on them.
// Here a workgroup of 4 waves owns (block, utterance, time range).  Per chunk of 16 time steps each
// thread fetches 8 consecutive time steps of ONE channel (lanes = consecutive channels: 128-byte
// segments), two chunks ahead, splits them once into fp16 hi | lo and stores 16 bytes per plane into a
// channel-major LDS stage [channel][16 t]; MFMA fragments (time = the MFMA K dimension) are then
// conflict-free 16-byte LDS reads shared by all waves.  Every fp32 product is the usual 3-product
// split on v_mfma_f32_32x32x16_f16 with fp32 accumulation; gradient operands are pre-scaled by an
// exact power of two from their running max-abs (see wn_gemm16.hip).
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 wl_h8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ void wl_scale_from_max(const float* slot, float& sc, float& inv) {
  sc = 1.0f;
  inv = 1.0f;
  if (!slot) return;
  const float m = *slot;
  if (m > 0.f && m < 3.0e38f) {
    int e;
    (void)frexpf(m, &e);
    e = max(-100, min(100, e));
    sc = ldexpf(1.0f, -e);
    inv = ldexpf(1.0f, e);
  }
}

}  // namespace

// C32 = R / 32 = D / 32 (1 or 2).  Channel order of a stage: x[t-d] (R) | x[t] (R) | du (2D) | z (D) | go (R);
// INNER: x[t-d] (R) | x[t] (R) | du (R)
template <int C32, bool INNER>
__global__ __launch_bounds__(256, 2) void wn_wgrad_layer_kernel(const WnWgLayer* layers, float* ws, float* slab,
                                                                int64_t P, int B, int T, int spb) {
  constexpr int R = 32 * C32, NCH = INNER ? 3 * R : 6 * R;
  constexpr int DUW = INNER ? R : 2 * R;                // channels (= row stride) of du
  constexpr int NXU = INNER ? 2 * 2 * R : 2 * 4 * R;    // stride-R units of a workgroup: (x[t-d] | x[t] (| z | go)) x two halves
  constexpr int NU = 1 + (NXU + 255) / 256;             // one du unit + the stride-R units of a thread
  constexpr int PLANE = NCH * 32;                       // bytes of one fp16 plane of a stage
  constexpr int STAGE = 2 * PLANE;
  // du tiles per wave.  INNER: C32 = 2: wave = x tile (tap, k tile), both du tiles; C32 = 1: waves 0, 2 own tap 0, 1
  constexpr int NJ = INNER ? C32 : ((C32 == 2) ? 4 : 1);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 2 * NCH * 4];
  float* bpart = reinterpret_cast<float*>(smem + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  const WnWgLayer Ld = layers[blockIdx.y];
  const int split = blockIdx.x;
  const int ub = split / spb, sp = split % spb;
  int len = (T + spb - 1) / spb;
  len = (len + 15) & ~15;
  const int r0 = sp * len, r1 = min(T, r0 + len);
  const int d = Ld.dilation;

  float su, inv_u, sh, inv_h;
  wl_scale_from_max(Ld.gmax_u_off >= 0 ? ws + Ld.gmax_u_off : nullptr, su, inv_u);
  wl_scale_from_max(Ld.gmax_h_off >= 0 ? ws + Ld.gmax_h_off : nullptr, sh, inv_h);

  // ---- this thread's units: (channel, half of the chunk).  Unit 0 is a du channel (row stride 2R);
  //      units 1.. are channels of x[t-d] | x[t] | z | go (row stride R: the plan guarantees ldz == R),
  //      so every unit's 8 loads are one base address plus compile-time offsets ----
  const float* uptr[NU];
  int ushift[NU], uch[NU], uhh[NU];
  float uscale[NU];
  bool uvalid[NU], ugrad[NU];
  {
    uvalid[0] = tid < 2 * DUW;
    const int c = uvalid[0] ? tid % DUW : 0;
    uch[0] = 2 * R + c;
    uhh[0] = uvalid[0] ? tid / DUW : 0;
    ushift[0] = 0;
    uscale[0] = su;
    ugrad[0] = true;
    uptr[0] = ws + Ld.du_off + (int64_t)ub * T * DUW + c;
  }
#pragma unroll
  for (int i = 1; i < NU; ++i) {
    const int uid = (i - 1) * 256 + tid;
    uvalid[i] = uid < NXU;
    const int c = uvalid[i] ? uid % (NXU / 2) : 0;
    uhh[i] = uvalid[i] ? uid / (NXU / 2) : 0;
    ushift[i] = 0;
    uscale[i] = 1.0f;
    ugrad[i] = false;
    if (c < 2 * R) {                                    // x[t - d] | x[t]
      uch[i] = c;
      uptr[i] = ws + Ld.x_off + (int64_t)ub * T * R + (c % R);
      ushift[i] = c < R ? d : 0;
    } else if (c < 3 * R) {                             // z
      uch[i] = 4 * R + (c - 2 * R);
      uptr[i] = ws + Ld.z_off + (int64_t)ub * T * R + (c - 2 * R);
    } else {                                            // go
      uch[i] = 5 * R + (c - 3 * R);
      uptr[i] = ws + Ld.go_off + (int64_t)ub * T * R + (c - 3 * R);
      uscale[i] = sh;
      ugrad[i] = true;
    }
  }
  float bsum[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) bsum[i] = 0.f;

  auto load_chunk = [&](int t0, float (&v)[NU][8]) {
    const bool interior = (t0 + 16 <= r1) && (t0 - d >= 0);          // workgroup-uniform
    wn_static_for<NU>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int LD = (i == 0) ? DUW : R;
      const int tb = t0 + 8 * uhh[i] - ushift[i];
      const float* p = uptr[i] + (int64_t)tb * LD;
      if (interior && uvalid[i]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = p[e * LD];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = uvalid[i] && (t0 + 8 * uhh[i] + e < r1) && (tb + e >= 0);
          v[i][e] = ok ? p[e * LD] : 0.f;
        }
      }
    });
  };
  auto store_chunk = [&](int stage, const float (&v)[NU][8]) {
    unsigned char* st = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if (!uvalid[i]) continue;
      wl_h8 hi, lo;
      float s8 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)(v[i][e] * uscale[i]);
        hi[e] = hh;
        lo[e] = (_Float16)__builtin_fmaf(v[i][e], uscale[i], -(float)hh);
        s8 += v[i][e];
      }
      if (ugrad[i]) bsum[i] += s8;
      *reinterpret_cast<wl_h8*>(st + uch[i] * 32 + uhh[i] * 16) = hi;
      *reinterpret_cast<wl_h8*>(st + PLANE + uch[i] * 32 + uhh[i] * 16) = lo;
    }
  };

  // ---- this wave's output tiles ----
  const int xt = (C32 == 2) ? wave : (wave >> 1);       // x tile (tap-major: tap = xt / C32, k tile = xt % C32)
  const int j0 = (C32 == 2 || INNER) ? 0 : (wave & 1);  // first du tile
  const bool has_d = !INNER || C32 == 2 || (wave & 1) == 0;   // owns dW_d tiles (INNER, C32 = 1: two tiles for four waves)
  const bool has_r = !INNER && ((C32 == 2) || wave == 0);     // owns a dW_r tile
  const int zt = (C32 == 2) ? (wave >> 1) : 0, gt = (C32 == 2) ? (wave & 1) : 0;
  f32x16 acc[NJ], accr;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accr[r] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* st = smem + stage * STAGE + tl * 32 + h * 16;
    auto frag = [&](int cb, wl_h8& hi, wl_h8& lo) {
      hi = *reinterpret_cast<const wl_h8*>(st + cb * 32);
      lo = *reinterpret_cast<const wl_h8*>(st + PLANE + cb * 32);
    };
    wl_h8 ah, al;
    if (has_d) frag(32 * xt, ah, al);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!has_d) break;
      wl_h8 bh, bl;
      frag(2 * R + 32 * (j0 + j), bh, bl);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
    }
    if (has_r) {
      wl_h8 zh, zl, gh, gl;
      frag(4 * R + 32 * zt, zh, zl);
      frag(5 * R + 32 * gt, gh, gl);
      accr = __builtin_amdgcn_mfma_f32_32x32x16_f16(zl, gh, accr, 0, 0, 0);
      accr = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, gl, accr, 0, 0, 0);
      accr = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, gh, accr, 0, 0, 0);
    }
  };

  // ---- pipeline: global loads two chunks ahead (registers), LDS one chunk ahead, one barrier per chunk ----
  float va[NU][8], vb[NU][8];
  if (r0 < r1) {
    load_chunk(r0, va);
    if (r0 + 16 < r1) load_chunk(r0 + 16, vb);
    store_chunk(0, va);
  }
  __syncthreads();
  int stage = 0;
  for (int t0 = r0; t0 < r1; t0 += 32) {
    // chunk t0 (stage `stage`); va is free, vb holds chunk t0 + 16
    if (t0 + 32 < r1) load_chunk(t0 + 32, va);
    compute(stage);
    if (t0 + 16 < r1) store_chunk(stage ^ 1, vb);
    __syncthreads();
    if (t0 + 16 >= r1) break;
    // chunk t0 + 16 (stage ^ 1); vb is free, va holds chunk t0 + 32
    if (t0 + 48 < r1) load_chunk(t0 + 48, vb);
    compute(stage ^ 1);
    if (t0 + 32 < r1) store_chunk(stage, va);
    __syncthreads();
  }

  // ---- partial results -> this split's slab row (laid out like the flat gradient buffer) ----
  float* row = slab + (int64_t)split * P;
  {
    const int tap = xt / C32, kt = xt % C32;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * kt + wn_drow(r, h);
        if (has_d) row[Ld.dwd_off + ((int64_t)tap * R + k) * DUW + 32 * (j0 + j) + tl] = acc[j][r] * inv_u;
      }
    if (has_r) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * zt + wn_drow(r, h);
        row[Ld.dwr_off + (int64_t)k * R + 32 * gt + tl] = accr[r] * inv_h;
      }
    }
  }
  // bias sums: the two halves of a chunk live in different threads
#pragma unroll
  for (int i = 0; i < NU; ++i)
    if (uvalid[i] && ugrad[i]) bpart[uhh[i] * NCH + uch[i]] = bsum[i];
  __syncthreads();
  for (int c = tid; c < NCH; c += 256) {
    if (c >= 2 * R && c < 2 * R + DUW) { if (Ld.dbd_off >= 0) row[Ld.dbd_off + (c - 2 * R)] = bpart[c] + bpart[NCH + c]; }
    else if (!INNER && c >= 5 * R) row[Ld.dbr_off + (c - 5 * R)] = bpart[c] + bpart[NCH + c];
  }
}

int wn_wgrad_layer_supported(int R, int D, int KS) { return R == D && (R == 32 || R == 64) && KS == 2; }

// inner: every entry is a non-gated conv of a deeper stack (x_off, du_off = its output gradient [rows][R], dwd_off, dbd_off,
// gmax_u_off, dilation; the 1x1 fields are ignored)
int wn_launch_wgrad_layers(const WnWgLayer* d_layers, int nlayers, int R, float* ws, float* slab, int64_t P, int B,
                           int T, int splits_per_b, hipStream_t s, int inner) {
  if (nlayers <= 0) return WN_OK;
  const dim3 grid((unsigned)(B * splits_per_b), (unsigned)nlayers);
  if (R == 64 && inner) hipLaunchKernelGGL((wn_wgrad_layer_kernel<2, true>), grid, dim3(256), 0, s, d_layers, ws, slab, P, B, T, splits_per_b);
  else if (R == 32 && inner) hipLaunchKernelGGL((wn_wgrad_layer_kernel<1, true>), grid, dim3(256), 0, s, d_layers, ws, slab, P, B, T, splits_per_b);
  else if (R == 64) hipLaunchKernelGGL((wn_wgrad_layer_kernel<2, false>), grid, dim3(256), 0, s, d_layers, ws, slab, P, B, T, splits_per_b);
  else if (R == 32) hipLaunchKernelGGL((wn_wgrad_layer_kernel<1, false>), grid, dim3(256), 0, s, d_layers, ws, slab, P, B, T, splits_per_b);
  else { wn_set_error("wgrad_layers: unsupported width %d", R); return WN_E_UNSUPPORTED; }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
