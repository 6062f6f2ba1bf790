// Generic contraction kernels for the WaveNet hot path on gfx950.
//
//  * wn_prep_*        : dense Keras-layout weights -> fragment-major A images
//  * wn_gemm_rows     : Y[t][n] = epi( sum_seg sum_k X_seg[t - shift_seg][k] * W_seg[k][n] )
//                       time on MFMA lanes, channels on MFMA rows; used for the input causal
//                       conv (src/model.py:84-88,228), the skip sum (model.py:236 folded with the
//                       per-block conv_skip, layers.py:216-217), the head 1x1 convs
//                       (model.py:105-119,237-238), the mapping Dense stack (model.py:141-148)
//                       and every backward-data product.
//  * wn_wgrad         : dW[k][n] = sum_t X[t - shift][k] * G[t][n]  (time is the MFMA K
//                       dimension), split over time into partial slabs + deterministic reduce.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

// ------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------
// fp16 hi/lo split image for v_mfma_f32_32x32x16_f16: per (k-step ks of 16, row tile j) two 1 KiB
// blocks (hi then lo); lane l holds 8 halfs, element jj <-> contraction index
// 16 ks + 8 (jj >> 2) + 4 (l >> 5) + (jj & 3)  -- the order in which a 32x32 accumulator tile's
// registers present their rows, so that tile can feed the next MFMA unchanged.
__device__ __forceinline__ void wn_prep16_body(const WnPrepDesc& d, const float* params, float* ws,
                                               int64_t start, int64_t stride) {
  const int JI = (d.I + 31) / 32;
  const int nks = (d.KK + 15) / 16;
  const int64_t total = (int64_t)nks * JI * 64 * 8;
  const float* src = params + d.src_off;
  __half* dst = reinterpret_cast<__half*>(ws + d.dst_off);
  for (int64_t idx = start; idx < total; idx += stride) {
    const int jj = idx & 7;
    const int lane = (idx >> 3) & 63;
    const int64_t blk = idx >> 9;
    const int j = blk % JI;
    const int ks = blk / JI;
    const int i = 32 * j + (lane & 31);
    const int kk = 16 * ks + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
    float v = 0.f;
    if (i < d.I && kk < d.KK) v = d.transpose ? src[(int64_t)kk * d.ld + i] : src[(int64_t)i * d.ld + kk];
    const __half hi = __float2half_rn(v);
    const __half lo = __float2half_rn(v - __half2float(hi));
    const int64_t base = (((int64_t)(ks + d.q_off) * d.JT + (j + d.j_off)) * 2) * 512 + lane * 8 + jj;
    dst[base] = hi;
    dst[base + 512] = lo;
  }
}

__device__ __forceinline__ void wn_prep_body(const WnPrepDesc& d, const float* params, float* ws,
                                             int64_t start, int64_t stride) {
  if (d.kind == 1) { wn_prep16_body(d, params, ws, start, stride); return; }
  const int JI = (d.I + 31) / 32;          // row tiles of this piece (== d.JT for whole images)
  const int nq = (d.KK + 7) / 8;
  const int64_t total = (int64_t)nq * JI * 256;
  const float* src = params + d.src_off;
  float* dst = ws + d.dst_off;
  for (int64_t idx = start; idx < total; idx += stride) {
    int e = idx & 3;
    int lane = (idx >> 2) & 63;
    int64_t blk = idx >> 8;
    int j = blk % JI;
    int q = blk / JI;
    int i = 32 * j + (lane & 31);
    int kk = 8 * q + 4 * (lane >> 5) + e;
    float v = 0.f;
    if (i < d.I && kk < d.KK) v = d.transpose ? src[(int64_t)kk * d.ld + i] : src[(int64_t)i * d.ld + kk];
    dst[(((int64_t)(q + d.q_off) * d.JT + (j + d.j_off)) * 64 + lane) * 4 + e] = v;
  }
}

__global__ void wn_prep_table_kernel(const WnPrepDesc* table, const float* params, float* ws) {
  WnPrepDesc d = table[blockIdx.y];
  wn_prep_body(d, params, ws, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
               (int64_t)gridDim.x * blockDim.x);
}

__global__ void wn_prep_one_kernel(WnPrepDesc d, const float* params, float* ws) {
  wn_prep_body(d, params, ws, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
               (int64_t)gridDim.x * blockDim.x);
}

// gx = workgroups per descriptor (16 suits images of up to ~64 k elements; the folded skip path's 245 k-element image wants more)
int wn_launch_prep_table(const WnPrepDesc* d_table, int n, const float* params, float* ws,
                         hipStream_t s, int gx) {
  if (n <= 0) return WN_OK;
  dim3 grid(gx > 0 ? gx : 16, n);
  hipLaunchKernelGGL(wn_prep_table_kernel, grid, dim3(256), 0, s, d_table, params, ws);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int wn_launch_prep_one(WnPrepDesc d, const float* params, float* ws, hipStream_t s) {
  hipLaunchKernelGGL(wn_prep_one_kernel, dim3(16), dim3(256), 0, s, d, params, ws);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_vecsum_kernel(WnVecSumArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.len) return;
  float acc = 0.f;
  for (int j = 0; j < a.count; ++j) acc += a.base[a.off0 + (int64_t)j * a.stride + i];
  a.out[i] = acc;
}

int wn_launch_vecsum(WnVecSumArgs a, hipStream_t s) {
  hipLaunchKernelGGL(wn_vecsum_kernel, dim3((a.len + 255) / 256), dim3(256), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// rows GEMM
// ------------------------------------------------------------------------------------------
template <int JT>
__global__ __launch_bounds__(256, (JT >= 8 ? 2 : (JT >= 4 ? 3 : 4))) void wn_gemm_rows_kernel(WnGemmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= (int64_t)a.B * tiles_per_b) return;   // wave-uniform; the kernel has no barriers
  const int b = (int)(tile / tiles_per_b);
  const int t0 = (int)(tile % tiles_per_b) * 32;
  const int tl = lane & 31, h = lane >> 5;
  const int t = t0 + tl;
  const int jb = blockIdx.y * JT;

  f32x16 acc[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // static segment indices: a runtime index into the by-value argument struct would make the
  // compiler copy it to private memory
#pragma unroll
  for (int s = 0; s < WN_MAXSEG; ++s) {
    if (s >= a.nseg) break;
    const float* sx = a.seg[s].x;
    const float* sfrag = a.seg[s].frag;
    const int sldx = a.seg[s].ldx, sK = a.seg[s].K, sshift = a.seg[s].shift, svec = a.seg[s].vec;
    const int splk = a.seg[s].plane_k;
    const int64_t splst = a.seg[s].plane_stride;
    const int ts = t - sshift;
    const bool valid = (t < a.T) && (ts >= 0) && (ts < a.T);
    const float* xrow = sx + ((int64_t)b * a.T + (valid ? ts : 0)) * sldx;
    const int nq = (sK + 7) >> 3;
    const f32x4* fr = reinterpret_cast<const f32x4*>(sfrag) + (int64_t)jb * 64 + lane;

    // operands of one k-quad: the lane's 4 activations and JT weight fragments
    auto load_x = [&](int q) -> f32x4 {
      const int k = 8 * q + 4 * h;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        // block-major operands: channel k lives in plane k / plane_k (plane_k is a multiple of 8)
        const float* xp = splk > 0 ? xrow + (int64_t)(k / splk) * splst + (k % splk) : xrow + k;
        if (svec && k + 3 < sK) {
          xv = *reinterpret_cast<const f32x4*>(xp);
        } else {
          if (k + 0 < sK) xv.x = xp[0];
          if (k + 1 < sK) xv.y = xp[1];
          if (k + 2 < sK) xv.z = xp[2];
          if (k + 3 < sK) xv.w = xp[3];
        }
      }
      return xv;
    };
    auto load_a = [&](int q, f32x4 (&av)[JT]) {
      wn_static_for<JT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (jb + j < a.JTtot) av[j] = fr[((int64_t)q * a.JTtot + j) * 64];   // block-uniform
      });
    };
    auto compute = [&](const f32x4& xv, const f32x4 (&av)[JT]) {
      wn_static_for<JT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (jb + j < a.JTtot) {
          acc[j] = wn_mfma(av[j].x, xv.x, acc[j]);
          acc[j] = wn_mfma(av[j].y, xv.y, acc[j]);
          acc[j] = wn_mfma(av[j].z, xv.z, acc[j]);
          acc[j] = wn_mfma(av[j].w, xv.w, acc[j]);
        }
      });
    };
    constexpr int PF = 8;
    if (JT == 1 && svec && splk == 0 && (sK & 7) == 0 && (nq % PF) == 0 && jb < a.JTtot) {   // wave-uniform
      // One row tile per wave: 4 dependent products a k-quad (256 clocks) hide nothing of a load's latency, and with a
      // few rows per launch (the last head layer of a queued generation step: 8 rows, K = 256, N = 30) there is no other
      // wave to hide it either.  The general pipeline below guards its loads (row validity, ragged K, image rows), and
      // hipcc drains vmcnt to 0 where such a guarded region ends: every k-quad then pays a whole round trip (17 us for
      // 32 quads).  This form has NO guarded load -- rows are clamped and zeroed by a select, the refill past the end
      // re-reads the last quad -- and a ring of PF register sets keeps PF - 1 quads in flight (17 -> 6 us).  Same
      // products in the same order.
      const f32x4* xp = reinterpret_cast<const f32x4*>(xrow + 4 * h);
      const f32x4* ap = fr;
      const int64_t astep = (int64_t)a.JTtot * 64;
      f32x4 xr[PF], ar[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) { xr[i] = xp[2 * i]; ar[i] = ap[i * astep]; }
      __builtin_amdgcn_sched_barrier(0);               // (or hipcc sinks every load next to its use, vmcnt(0) behind it)
      for (int q0 = 0; q0 < nq; q0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          f32x4 xv = xr[i];
          if (!valid) xv = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 av = ar[i];
          acc[0] = wn_mfma(av.x, xv.x, acc[0]);
          acc[0] = wn_mfma(av.y, xv.y, acc[0]);
          acc[0] = wn_mfma(av.z, xv.z, acc[0]);
          acc[0] = wn_mfma(av.w, xv.w, acc[0]);
          const int qn = min(q0 + i + PF, nq - 1);
          xr[i] = xp[2 * qn];
          ar[i] = ap[qn * astep];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    // software pipeline, one k-quad ahead, two named register sets (no copies)
    f32x4 x0, x1, a0[JT], a1[JT];
    x0 = load_x(0);
    load_a(0, a0);
    int q = 0;
    for (; q + 2 <= nq; q += 2) {
      x1 = load_x(q + 1);
      load_a(q + 1, a1);
      compute(x0, a0);
      if (q + 2 < nq) {
        x0 = load_x(q + 2);
        load_a(q + 2, a0);
      }
      compute(x1, a1);
    }
    if (q < nq) compute(x0, a0);
    }
  }

  if (t >= a.T) return;
  const int64_t row = (int64_t)b * a.T + t;
  wn_static_for<JT * 4>([&](auto jc) {
    constexpr int j = decltype(jc)::value / 4;
    constexpr int rq = decltype(jc)::value % 4;
    {
      const int n0 = 32 * (jb + j) + 8 * rq + 4 * h;
      if (n0 >= a.N) return;
      float v[4];
      v[0] = acc[j][4 * rq + 0]; v[1] = acc[j][4 * rq + 1];
      v[2] = acc[j][4 * rq + 2]; v[3] = acc[j][4 * rq + 3];
      const bool full = a.vec_out && (n0 + 3 < a.N);
      if (full) {
        if (a.bias) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n0);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (a.rowbias) {
          const float* rb = a.rowbias + (int64_t)b * a.ld_rowbias + n0;
          v[0] += rb[0]; v[1] += rb[1]; v[2] += rb[2]; v[3] += rb[3];
        }
        if (a.addc) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(a.addc + row * a.ld_addc + n0);
          v[0] += cv.x; v[1] += cv.y; v[2] += cv.z; v[3] += cv.w;
        }
        if (a.epi == WN_EPI_PLAIN) {
          f32x4 o;
          o.x = wn_act(v[0], a.act); o.y = wn_act(v[1], a.act);
          o.z = wn_act(v[2], a.act); o.w = wn_act(v[3], a.act);
          *reinterpret_cast<f32x4*>(a.y + row * a.ldy + n0) = o;
        } else if (a.epi == WN_EPI_DACT) {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(a.aux + row * a.ld_aux + n0);
          f32x4 o;
          o.x = v[0] * wn_dact_from_y(yv.x, a.act); o.y = v[1] * wn_dact_from_y(yv.y, a.act);
          o.z = v[2] * wn_dact_from_y(yv.z, a.act); o.w = v[3] * wn_dact_from_y(yv.w, a.act);
          *reinterpret_cast<f32x4*>(a.y + row * a.ldy + n0) = o;
        } else {  // WN_EPI_GATE_BWD
          const f32x4 gv = *reinterpret_cast<const f32x4*>(a.aux + row * a.ld_aux + n0);
          const f32x4 zv = *reinterpret_cast<const f32x4*>(a.aux2 + row * a.ld_aux2 + n0);
          float f4[4], g4[4];
          wn_gate_bwd(v[0], gv.x, zv.x, f4[0], g4[0]);
          wn_gate_bwd(v[1], gv.y, zv.y, f4[1], g4[1]);
          wn_gate_bwd(v[2], gv.z, zv.z, f4[2], g4[2]);
          wn_gate_bwd(v[3], gv.w, zv.w, f4[3], g4[3]);
          const f32x4 of = {f4[0], f4[1], f4[2], f4[3]}, og = {g4[0], g4[1], g4[2], g4[3]};
          *reinterpret_cast<f32x4*>(a.y + row * a.ldy + n0) = of;
          *reinterpret_cast<f32x4*>(a.y + row * a.ldy + a.N + n0) = og;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n >= a.N) continue;
          float w = v[e];
          if (a.bias) w += a.bias[n];
          if (a.rowbias) w += a.rowbias[(int64_t)b * a.ld_rowbias + n];
          if (a.addc) w += a.addc[row * a.ld_addc + n];
          if (a.epi == WN_EPI_PLAIN) {
            a.y[row * a.ldy + n] = wn_act(w, a.act);
          } else if (a.epi == WN_EPI_DACT) {
            a.y[row * a.ldy + n] = w * wn_dact_from_y(a.aux[row * a.ld_aux + n], a.act);
          } else {
            float duf, dug;
            wn_gate_bwd(w, a.aux[row * a.ld_aux + n], a.aux2[row * a.ld_aux2 + n], duf, dug);
            a.y[row * a.ldy + n] = duf;
            a.y[row * a.ldy + a.N + n] = dug;
          }
        }
      }
    }
  });
}

int wn_launch_gemm_rows(const WnGemmArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.T <= 0 || a.N <= 0) return WN_OK;
  if (a.nseg < 1 || a.nseg > WN_MAXSEG) { wn_set_error("gemm_rows: bad nseg %d", a.nseg); return WN_E_INVALID; }
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  const int64_t gx = (tiles + 3) / 4;
  if (gx > 0x7fffffffLL) { wn_set_error("gemm_rows: grid too large"); return WN_E_UNSUPPORTED; }
  const int jt_need = (a.N + 31) / 32;
  if (jt_need > a.JTtot) { wn_set_error("gemm_rows: N %d exceeds image rows %d", a.N, a.JTtot * 32); return WN_E_INVALID; }
  // pick the register blocking: as many row tiles per wave as fit two waves per SIMD
  if (jt_need <= 1) {
    hipLaunchKernelGGL(wn_gemm_rows_kernel<1>, dim3((unsigned)gx, 1), dim3(256), 0, s, a);
  } else if (jt_need <= 2) {
    hipLaunchKernelGGL(wn_gemm_rows_kernel<2>, dim3((unsigned)gx, 1), dim3(256), 0, s, a);
  } else if (jt_need <= 4) {
    hipLaunchKernelGGL(wn_gemm_rows_kernel<4>, dim3((unsigned)gx, 1), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(wn_gemm_rows_kernel<8>, dim3((unsigned)gx, (jt_need + 7) / 8), dim3(256), 0, s, a);
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// weight-gradient GEMM: channels on lanes, time is the contraction
// ------------------------------------------------------------------------------------------
#define WG_TM 2
#define WG_TN 4

// one wave: rows [r0, r1) of utterance b, output block (k0.., n0..) of WG_TM x WG_TN tiles
struct WnWgUnit {
  const float* x; int ldx; int K; int shift;
  const float* g; int ldg; int N;
  int T; int b; int r0; int r1; int k0; int n0;
  float* out; int out_ld;         // dW block origin is out[k * out_ld + n]
  float* bias;                    // db[n] or null
};

__device__ __forceinline__ void wn_wgrad_body(const WnWgUnit& a) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & 31, h = lane >> 5;
  const int k0 = a.k0, n0 = a.n0, r0 = a.r0, r1 = a.r1;

  f32x16 acc[WG_TM][WG_TN];
#pragma unroll
  for (int i = 0; i < WG_TM; ++i)
#pragma unroll
    for (int j = 0; j < WG_TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[WG_TN];
#pragma unroll
  for (int j = 0; j < WG_TN; ++j) bsum[j] = 0.f;

  const float* xb = a.x + (int64_t)a.b * a.T * a.ldx;
  const float* gb = a.g + (int64_t)a.b * a.T * a.ldg;
  // WG_CH time-step pairs per chunk; the next chunk's operands are loaded before the current
  // chunk's MFMAs (two named register sets)
  constexpr int WG_CH = 4;
  // running row pointers (advanced by whole chunks) + channel masks hoisted out of the loop
  const int64_t xstep = (int64_t)2 * a.ldx, gstep = (int64_t)2 * a.ldg;
  bool kin[WG_TM], nin[WG_TN];
#pragma unroll
  for (int i = 0; i < WG_TM; ++i) kin[i] = (k0 + 32 * i + tl) < a.K;
#pragma unroll
  for (int j = 0; j < WG_TN; ++j) nin[j] = (n0 + 32 * j + tl) < a.N;
  const float* xbase = xb + k0 + tl;
  const float* gbase = gb + n0 + tl;
  auto load_chunk = [&](int tt0, float (&av)[WG_CH][WG_TM], float (&bv)[WG_CH][WG_TN]) {
    const float* px = xbase + (int64_t)(tt0 + h - a.shift) * a.ldx;
    const float* pg = gbase + (int64_t)(tt0 + h) * a.ldg;
#pragma unroll
    for (int c = 0; c < WG_CH; ++c) {
      const int t = tt0 + 2 * c + h;
      const int ts = t - a.shift;
      const bool tv = t < r1;
      const bool xv = tv && ts >= 0 && ts < a.T;
#pragma unroll
      for (int i = 0; i < WG_TM; ++i) av[c][i] = (xv && kin[i]) ? px[c * xstep + 32 * i] : 0.f;
#pragma unroll
      for (int j = 0; j < WG_TN; ++j) bv[c][j] = (tv && nin[j]) ? pg[c * gstep + 32 * j] : 0.f;
    }
  };
  auto compute_chunk = [&](const float (&av)[WG_CH][WG_TM], const float (&bv)[WG_CH][WG_TN]) {
#pragma unroll
    for (int c = 0; c < WG_CH; ++c) {
#pragma unroll
      for (int j = 0; j < WG_TN; ++j) bsum[j] += bv[c][j];
#pragma unroll
      for (int i = 0; i < WG_TM; ++i)
#pragma unroll
        for (int j = 0; j < WG_TN; ++j) acc[i][j] = wn_mfma(av[c][i], bv[c][j], acc[i][j]);
    }
  };
  {
    float av0[WG_CH][WG_TM], bv0[WG_CH][WG_TN], av1[WG_CH][WG_TM], bv1[WG_CH][WG_TN];
    constexpr int STEP = 2 * WG_CH;
    int tt = r0;
    if (tt < r1) load_chunk(tt, av0, bv0);
    for (; tt < r1; tt += 2 * STEP) {
      if (tt + STEP < r1) load_chunk(tt + STEP, av1, bv1);
      compute_chunk(av0, bv0);
      if (tt + STEP < r1) {
        if (tt + 2 * STEP < r1) load_chunk(tt + 2 * STEP, av0, bv0);
        compute_chunk(av1, bv1);
      }
    }
  }

#pragma unroll
  for (int i = 0; i < WG_TM; ++i)
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const int n = n0 + 32 * j + tl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + 32 * i + wn_drow(r, h);
        if (k < a.K && n < a.N) a.out[(int64_t)k * a.out_ld + n] = acc[i][j][r];
      }
    }
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const float tot = bsum[j] + __shfl_xor(bsum[j], 32);
      const int n = n0 + 32 * j + tl;
      if (h == 0 && n < a.N) a.bias[n] = tot;
    }
  }
}

// Split-precision form of the same unit (fp16 hi/lo, 3 products, fp32 accumulate) on
// v_mfma_f32_32x32x16_f16 with time as the MFMA K dimension: lane (c = l&31, h = l>>5) holds 8
// consecutive time steps t0 + 8h .. +7 of its channel for each operand tile; loads stay one dword per
// lane, 128 contiguous bytes per half wave.  gsc / inv_gsc: exact power-of-two scaling of the
// gradient operand (running max-abs of its tensor), undone on the accumulators.
typedef _Float16 wg_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void wn_wgrad_body16(const WnWgUnit& a, float gsc, float inv_gsc) {
  const int lane = threadIdx.x & 63;
  const int tl = lane & 31, h = lane >> 5;
  const int k0 = a.k0, n0 = a.n0, r0 = a.r0, r1 = a.r1;

  f32x16 acc[WG_TM][WG_TN];
#pragma unroll
  for (int i = 0; i < WG_TM; ++i)
#pragma unroll
    for (int j = 0; j < WG_TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[WG_TN];
#pragma unroll
  for (int j = 0; j < WG_TN; ++j) bsum[j] = 0.f;

  bool kin[WG_TM], nin[WG_TN];
#pragma unroll
  for (int i = 0; i < WG_TM; ++i) kin[i] = (k0 + 32 * i + tl) < a.K;
#pragma unroll
  for (int j = 0; j < WG_TN; ++j) nin[j] = (n0 + 32 * j + tl) < a.N;
  const float* xbase = a.x + (int64_t)a.b * a.T * a.ldx + k0 + tl;
  const float* gbase = a.g + (int64_t)a.b * a.T * a.ldg + n0 + tl;

  // one chunk = 16 time steps; raw fp32 operands of the next chunk are in flight during the MFMAs.
  // Chunks that lie wholly inside [r0, r1) and whose shifted rows lie inside the utterance take a
  // mask-free path (wave-uniform test): only channel masks of ragged widths remain.
  bool kall = true, nall = true;
#pragma unroll
  for (int i = 0; i < WG_TM; ++i) kall = kall && (k0 + 32 * i + 31 < a.K);
#pragma unroll
  for (int j = 0; j < WG_TN; ++j) nall = nall && (n0 + 32 * j + 31 < a.N);
  const bool chan_full = kall && nall;                       // wave-uniform
  auto load_chunk = [&](int tt0, float (&av)[WG_TM][8], float (&bv)[WG_TN][8]) {
    const int tb = tt0 + 8 * h;
    const float* px = xbase + (int64_t)(tb - a.shift) * a.ldx;
    const float* pg = gbase + (int64_t)tb * a.ldg;
    const bool interior = chan_full && (tt0 + 16 <= r1) && (tt0 - a.shift >= 0) && (tt0 + 16 - a.shift <= a.T);
    if (interior) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int i = 0; i < WG_TM; ++i) av[i][e] = px[(int64_t)e * a.ldx + 32 * i];
#pragma unroll
        for (int j = 0; j < WG_TN; ++j) bv[j][e] = pg[(int64_t)e * a.ldg + 32 * j];
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = tb + e;
      const int ts = t - a.shift;
      const bool tv = t < r1;
      const bool xv = tv && ts >= 0 && ts < a.T;
#pragma unroll
      for (int i = 0; i < WG_TM; ++i) av[i][e] = (xv && kin[i]) ? px[(int64_t)e * a.ldx + 32 * i] : 0.f;
#pragma unroll
      for (int j = 0; j < WG_TN; ++j) bv[j][e] = (tv && nin[j]) ? pg[(int64_t)e * a.ldg + 32 * j] : 0.f;
    }
  };
  // hi = rn(v), lo = rn(v - hi): round-to-nearest keeps the split error at 2^-22 |v| and unbiased (a
  // packed round-toward-zero split is ~0.15 ms per step faster but its truncation error is visible
  // after Adam's normalisation on near-zero gradient entries)
  auto split8 = [&](const float (&v)[8], wg_h8& hi, wg_h8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 hh = (_Float16)v[e];
      hi[e] = hh;
      lo[e] = (_Float16)(v[e] - (float)hh);
    }
  };
  auto compute_chunk = [&](const float (&av)[WG_TM][8], const float (&bv)[WG_TN][8]) {
    wg_h8 ah[WG_TM], al[WG_TM];
#pragma unroll
    for (int i = 0; i < WG_TM; ++i) split8(av[i], ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      wg_h8 bh, bl;
      float sv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bsum[j] += bv[j][e];
        sv[e] = bv[j][e] * gsc;
      }
      split8(sv, bh, bl);
#pragma unroll
      for (int i = 0; i < WG_TM; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
    }
  };
  {
    float av0[WG_TM][8], bv0[WG_TN][8], av1[WG_TM][8], bv1[WG_TN][8];
    int tt = r0;
    if (tt < r1) load_chunk(tt, av0, bv0);
    for (; tt < r1; tt += 32) {
      if (tt + 16 < r1) load_chunk(tt + 16, av1, bv1);
      compute_chunk(av0, bv0);
      if (tt + 16 < r1) {
        if (tt + 32 < r1) load_chunk(tt + 32, av0, bv0);
        compute_chunk(av1, bv1);
      }
    }
  }

#pragma unroll
  for (int i = 0; i < WG_TM; ++i)
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const int n = n0 + 32 * j + tl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + 32 * i + wn_drow(r, h);
        if (k < a.K && n < a.N) a.out[(int64_t)k * a.out_ld + n] = acc[i][j][r] * inv_gsc;
      }
    }
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const float tot = bsum[j] + __shfl_xor(bsum[j], 32);
      const int n = n0 + 32 * j + tl;
      if (h == 0 && n < a.N) a.bias[n] = tot;
    }
  }
}

__global__ __launch_bounds__(256, 2) void wn_wgrad_kernel(WnWgradArgs a, int nkb, int nnb) {
  const int wave = threadIdx.x >> 6;
  const int nsplit = a.B * a.splits_per_b;
  const int split = blockIdx.x * 4 + wave;
  if (split >= nsplit) return;                         // wave-uniform
  const int kb = blockIdx.y / nnb, nb = blockIdx.y % nnb;
  const int sp = split % a.splits_per_b;
  int len = (a.T + a.splits_per_b - 1) / a.splits_per_b;
  len = (len + 1) & ~1;
  WnWgUnit u;
  u.x = a.x; u.ldx = a.ldx; u.K = a.K; u.shift = a.shift; u.g = a.g; u.ldg = a.ldg; u.N = a.N;
  u.T = a.T; u.b = split / a.splits_per_b; u.r0 = sp * len; u.r1 = min(a.T, u.r0 + len);
  u.k0 = kb * 32 * WG_TM; u.n0 = nb * 32 * WG_TN;
  u.out = a.slab + (int64_t)split * a.K * a.N; u.out_ld = a.N;
  u.bias = (a.slab_bias && kb == 0) ? a.slab_bias + (int64_t)split * a.N : nullptr;
  wn_wgrad_body(u);
}

// Batched form: a table of jobs (one per K-block x N-block of some dW), every job split over
// time; partial results go to slab[split][P] laid out exactly like the flat gradient buffer.
template <bool F16>
__global__ __launch_bounds__(256, 2) void wn_wgrad_batched_kernel(const WnWgJob* jobs, float* ws, float* slab,
                                                                  int64_t P, int B, int T, int splits_per_b) {
  const int wave = threadIdx.x >> 6;
  const int nsplit = B * splits_per_b;
  const int split = blockIdx.x * 4 + wave;
  if (split >= nsplit) return;                         // wave-uniform
  const WnWgJob j = jobs[blockIdx.y];
  const int sp = split % splits_per_b;
  int len = (T + splits_per_b - 1) / splits_per_b;
  len = (len + 1) & ~1;
  WnWgUnit u;
  u.x = ws + j.x_off; u.ldx = j.ldx; u.K = j.K; u.shift = j.shift; u.g = ws + j.g_off; u.ldg = j.ldg; u.N = j.N;
  u.T = T; u.b = split / splits_per_b; u.r0 = sp * len; u.r1 = min(T, u.r0 + len);
  u.k0 = j.k0; u.n0 = j.n0;
  float* row = slab + (int64_t)split * P;
  u.out = row + j.out_off; u.out_ld = j.N;
  u.bias = j.bias_off >= 0 ? row + j.bias_off : nullptr;
  if constexpr (F16) {
    float gsc = 1.0f, inv = 1.0f;
    if (j.gmax_off >= 0) {
      const float m = ws[j.gmax_off];
      if (m > 0.f && m < 3.0e38f) {
        int e;
        (void)frexpf(m, &e);
        e = max(-100, min(100, e));
        gsc = ldexpf(1.0f, -e);
        inv = ldexpf(1.0f, e);
      }
    }
    wn_wgrad_body16(u, gsc, inv);
  } else {
    wn_wgrad_body(u);
  }
}

int wn_launch_wgrad_batched(const WnWgJob* d_jobs, int njobs, float* ws, float* slab, int64_t P, int B, int T,
                            int splits_per_b, hipStream_t s, bool exact_fp32) {
  if (njobs <= 0) return WN_OK;
  const int nsplit = B * splits_per_b;
  // knob 1 = 1 forces exact-fp32 MFMA; knob 3 = 1 keeps the weight gradients alone on fp32; exact_fp32: gradient operands
  // without a max-abs slot (inner convs of stacks deeper than 1)
  if (exact_fp32 || wn_debug_get(1) == 1 || wn_debug_get(3) == 1)
    hipLaunchKernelGGL(wn_wgrad_batched_kernel<false>, dim3((nsplit + 3) / 4, njobs), dim3(256), 0, s, d_jobs, ws, slab, P,
                       B, T, splits_per_b);
  else
    hipLaunchKernelGGL(wn_wgrad_batched_kernel<true>, dim3((nsplit + 3) / 4, njobs), dim3(256), 0, s, d_jobs, ws, slab, P,
                       B, T, splits_per_b);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// grads[off + i] = sum_s slab[s][off + i] over the tensors of a table
__global__ void wn_reduce_table_kernel(const float* slab, int nsplit, int64_t P, float* out,
                                       const WnTensorDesc* table) {
  const WnTensorDesc d = table[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.len;
       i += (int64_t)gridDim.x * blockDim.x) {
    // 8 rows requested at a time, added in row order (the sum is the same as one by one)
    float acc = 0.f;
    const float* col = slab + d.off + i;
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = col[(int64_t)(s + e) * P];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += v[e];
    }
    for (; s < nsplit; ++s) acc += col[(int64_t)s * P];
    out[d.off + i] = acc;
  }
}

// The same sums with one element per thread and exactly as many workgroups as the tensors need: the grid above is
// (64, tensors) whatever their lengths -- 11 520 workgroups at configs[1], most of which find nothing to do (42 us), and
// 15 serial elements per thread for the one long tensor of the folded skip path.  bf.v[t] = first workgroup of tensor t.
struct WnBlkFirst { int v[257]; };
__global__ __launch_bounds__(256) void wn_reduce_flat_kernel(const float* slab, int nsplit, int64_t P, float* out,
                                                             const WnTensorDesc* table, WnBlkFirst bf, int n) {
  int lo = 0, hi = n;                                   // largest t with bf.v[t] <= blockIdx.x (workgroup-uniform)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (bf.v[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const WnTensorDesc d = table[lo];
  const int64_t i = (int64_t)((int)blockIdx.x - bf.v[lo]) * 256 + threadIdx.x;
  if (i >= d.len) return;
  // rows requested 16 at a time, added in row order (the sum is the same as one by one)
  float acc = 0.f;
  const float* col = slab + d.off + i;
  int s = 0;
  for (; s + 16 <= nsplit; s += 16) {
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = col[(int64_t)(s + e) * P];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += v[e];
  }
  for (; s < nsplit; ++s) acc += col[(int64_t)s * P];
  out[d.off + i] = acc;
}

// h_table: host copy of the same n descriptors (or null: the (64, n) grid)
int wn_launch_reduce_table(const float* slab, int nsplit, int64_t P, float* out, const WnTensorDesc* d_table,
                           int n, hipStream_t s, const WnTensorDesc* h_table) {
  if (n <= 0) return WN_OK;
  if (h_table && n <= 256) {
    WnBlkFirst bf;
    int64_t nblk = 0;
    for (int t = 0; t < n; ++t) { bf.v[t] = (int)nblk; nblk += (h_table[t].len + 255) / 256; }
    for (int t = n; t <= 256; ++t) bf.v[t] = (int)nblk;
    if (nblk <= 0) return WN_OK;
    if (nblk < (int64_t)1 << 30) {
      hipLaunchKernelGGL(wn_reduce_flat_kernel, dim3((unsigned)nblk), dim3(256), 0, s, slab, nsplit, P, out, d_table, bf, n);
      WN_HIP_CHECK(hipGetLastError());
      return WN_OK;
    }
  }
  hipLaunchKernelGGL(wn_reduce_table_kernel, dim3(64, n), dim3(256), 0, s, slab, nsplit, P, out, d_table);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int wn_wgrad_tile_k() { return 32 * WG_TM; }
int wn_wgrad_tile_n() { return 32 * WG_TN; }

int wn_wgrad_choose_splits(int B, int T, int K, int N) {
  const int jobs = ((K + 32 * WG_TM - 1) / (32 * WG_TM)) * ((N + 32 * WG_TN - 1) / (32 * WG_TN));
  // aim at one wave per SIMD over the chip (1024 waves), at least 256 rows per split: every
  // split costs a K x N partial slab that has to be written and reduced
  int want = (1024 + jobs - 1) / jobs;
  int per_b = (want + B - 1) / B;
  const int max_per_b = (T + 255) / 256;
  if (per_b > max_per_b) per_b = max_per_b;
  if (per_b < 1) per_b = 1;
  return per_b;
}

int wn_launch_wgrad(const WnWgradArgs& a, hipStream_t s) {
  if (a.K <= 0 || a.N <= 0 || a.B <= 0 || a.T <= 0) return WN_OK;
  const int nkb = (a.K + 32 * WG_TM - 1) / (32 * WG_TM);
  const int nnb = (a.N + 32 * WG_TN - 1) / (32 * WG_TN);
  const int nsplit = a.B * a.splits_per_b;
  dim3 grid((nsplit + 3) / 4, nkb * nnb);
  hipLaunchKernelGGL(wn_wgrad_kernel, grid, dim3(256), 0, s, a, nkb, nnb);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// Stage A: fold groups of WN_RED_GROUP splits in place (row g*GROUP of the slab receives the
// group's sum; only this thread touches column i of those rows).  Stage B: sum the group rows
// in a fixed order and scatter to the gradient tensor.  Both are deterministic.
#define WN_RED_GROUP 16
__global__ void wn_reduce_stageA_kernel(float* slab, int nsplit, int64_t total) {
  const int g = blockIdx.y;
  const int s0 = g * WN_RED_GROUP;
  const int s1 = min(nsplit, s0 + WN_RED_GROUP);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = s0; s < s1; ++s) acc += slab[(int64_t)s * total + i];
    slab[(int64_t)s0 * total + i] = acc;
  }
}

__global__ void wn_reduce_kernel(WnReduceArgs a, int stride) {
  const int64_t total = (int64_t)a.K * a.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < a.nsplit; s += stride) acc += a.slab[(int64_t)s * total + i];
    const int k = (int)(i / a.N), n = (int)(i % a.N);
    float* dst = a.out + (int64_t)(k / a.seg_len) * a.seg_stride + (int64_t)(k % a.seg_len) * a.N + n;
    for (int r = 0; r < a.replicate; ++r) {
      float* d = dst + (int64_t)r * a.rep_stride;
      *d = a.accumulate ? (*d + acc) : acc;
    }
  }
}

int wn_launch_reduce(const WnReduceArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.K * a.N;
  if (total <= 0) return WN_OK;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  int stride = 1;
  if (a.nsplit > 2 * WN_RED_GROUP) {
    const int groups = (a.nsplit + WN_RED_GROUP - 1) / WN_RED_GROUP;
    hipLaunchKernelGGL(wn_reduce_stageA_kernel, dim3(blocks, groups), dim3(256), 0, s,
                       const_cast<float*>(a.slab), a.nsplit, total);
    stride = WN_RED_GROUP;
  }
  hipLaunchKernelGGL(wn_reduce_kernel, dim3(blocks), dim3(256), 0, s, a, stride);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
