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
#include "wn_kernels.h"

// ------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wn_prep_body(const WnPrepDesc& d, const float* params, float* ws,
                                             int64_t start, int64_t stride) {
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

int wn_launch_prep_table(const WnPrepDesc* d_table, int n, const float* params, float* ws,
                         hipStream_t s) {
  if (n <= 0) return WN_OK;
  dim3 grid(16, n);
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
__global__ __launch_bounds__(256) void wn_gemm_rows_kernel(WnGemmArgs a) {
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
    const int ts = t - sshift;
    const bool valid = (t < a.T) && (ts >= 0) && (ts < a.T);
    const float* xrow = sx + ((int64_t)b * a.T + (valid ? ts : 0)) * sldx;
    const int nq = (sK + 7) >> 3;
    const f32x4* fr = reinterpret_cast<const f32x4*>(sfrag) + (int64_t)jb * 64 + lane;
    for (int q = 0; q < nq; ++q) {
      const int k = 8 * q + 4 * h;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        if (svec && k + 3 < sK) {
          xv = *reinterpret_cast<const f32x4*>(xrow + k);
        } else {
          if (k + 0 < sK) xv.x = xrow[k + 0];
          if (k + 1 < sK) xv.y = xrow[k + 1];
          if (k + 2 < sK) xv.z = xrow[k + 2];
          if (k + 3 < sK) xv.w = xrow[k + 3];
        }
      }
      wn_static_for<JT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (jb + j < a.JTtot) {   // block-uniform
          const f32x4 av = fr[((int64_t)q * a.JTtot + j) * 64];
          acc[j] = wn_mfma(av.x, xv.x, acc[j]);
          acc[j] = wn_mfma(av.y, xv.y, acc[j]);
          acc[j] = wn_mfma(av.z, xv.z, acc[j]);
          acc[j] = wn_mfma(av.w, xv.w, acc[j]);
        }
      });
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
          const f32x4 av = *reinterpret_cast<const f32x4*>(a.aux + row * a.ld_aux + n0);
          const f32x4 gv = *reinterpret_cast<const f32x4*>(a.aux + row * a.ld_aux + a.N + n0);
          f32x4 of, og;
          of.x = v[0] * gv.x * (1.f - av.x * av.x); og.x = v[0] * av.x * gv.x * (1.f - gv.x);
          of.y = v[1] * gv.y * (1.f - av.y * av.y); og.y = v[1] * av.y * gv.y * (1.f - gv.y);
          of.z = v[2] * gv.z * (1.f - av.z * av.z); og.z = v[2] * av.z * gv.z * (1.f - gv.z);
          of.w = v[3] * gv.w * (1.f - av.w * av.w); og.w = v[3] * av.w * gv.w * (1.f - gv.w);
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
            const float av = a.aux[row * a.ld_aux + n];
            const float gv = a.aux[row * a.ld_aux + a.N + n];
            a.y[row * a.ldy + n] = w * gv * (1.f - av * av);
            a.y[row * a.ldy + a.N + n] = w * av * gv * (1.f - gv);
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

__global__ __launch_bounds__(256) void wn_wgrad_kernel(WnWgradArgs a, int nkb, int nnb) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tl = lane & 31, h = lane >> 5;
  const int nsplit = a.B * a.splits_per_b;
  const int split = blockIdx.x * 4 + wave;
  if (split >= nsplit) return;                         // wave-uniform
  const int kb = blockIdx.y / nnb, nb = blockIdx.y % nnb;
  const int k0 = kb * 32 * WG_TM, n0 = nb * 32 * WG_TN;
  const int b = split / a.splits_per_b;
  const int sp = split % a.splits_per_b;
  int len = (a.T + a.splits_per_b - 1) / a.splits_per_b;
  len = (len + 1) & ~1;
  const int r0 = sp * len;
  const int r1 = min(a.T, r0 + len);

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

  const float* xb = a.x + (int64_t)b * a.T * a.ldx;
  const float* gb = a.g + (int64_t)b * a.T * a.ldg;
  for (int tt = r0; tt < r1; tt += 2) {
    const int t = tt + h;
    const int ts = t - a.shift;
    const bool tv = t < r1;
    const bool xv = tv && ts >= 0 && ts < a.T;
    float av[WG_TM], bv[WG_TN];
#pragma unroll
    for (int i = 0; i < WG_TM; ++i) {
      const int c = k0 + 32 * i + tl;
      av[i] = (xv && c < a.K) ? xb[(int64_t)ts * a.ldx + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const int n = n0 + 32 * j + tl;
      bv[j] = (tv && n < a.N) ? gb[(int64_t)t * a.ldg + n] : 0.f;
      bsum[j] += bv[j];
    }
#pragma unroll
    for (int i = 0; i < WG_TM; ++i)
#pragma unroll
      for (int j = 0; j < WG_TN; ++j) acc[i][j] = wn_mfma(av[i], bv[j], acc[i][j]);
  }

  float* slab = a.slab + (int64_t)split * a.K * a.N;
#pragma unroll
  for (int i = 0; i < WG_TM; ++i)
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const int n = n0 + 32 * j + tl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + 32 * i + wn_drow(r, h);
        if (k < a.K && n < a.N) slab[(int64_t)k * a.N + n] = acc[i][j][r];
      }
    }
  if (a.slab_bias && kb == 0) {
#pragma unroll
    for (int j = 0; j < WG_TN; ++j) {
      const float tot = bsum[j] + __shfl_xor(bsum[j], 32);
      const int n = n0 + 32 * j + tl;
      if (h == 0 && n < a.N) a.slab_bias[(int64_t)split * a.N + n] = tot;
    }
  }
}

int wn_wgrad_choose_splits(int B, int T, int K, int N) {
  const int jobs = ((K + 32 * WG_TM - 1) / (32 * WG_TM)) * ((N + 32 * WG_TN - 1) / (32 * WG_TN));
  // aim at ~2 waves per SIMD over the chip (2048 waves), at least 64 rows per split
  int want = (2048 + jobs - 1) / jobs;
  int per_b = (want + B - 1) / B;
  const int max_per_b = (T + 63) / 64;
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

__global__ void wn_reduce_kernel(WnReduceArgs a) {
  const int64_t total = (int64_t)a.K * a.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < a.nsplit; ++s) acc += a.slab[(int64_t)s * total + i];
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
  hipLaunchKernelGGL(wn_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
