// Fused WaveNet residual block forward for gfx950 (reference: WaveNetLayer.call,
// src/layers.py:178-224, depth-1 dilated stack).
//
// One wave owns 32 consecutive time steps of one utterance (time on the MFMA lanes) and
// carries them through the whole block without leaving registers:
//
//   u[2D][t]  = b_d (+ cb[b]) + sum_tap W_d[tap]^T x[t - (KS-1-tap) d]     (MFMA, K = KS*R)
//   z[D][t]   = tanh(u[:D]) * sigmoid(u[D:])                               (VALU, same lane)
//   o[R][t]   = b_r + W_r^T z                                              (MFMA, K = D; the
//               D tile registers of z ARE the B operand, no LDS / shuffle)
//   x_out     = o + x[t]   (the last tap's B operand registers are x[t] in D layout)
//
// Filter channel c and gate channel c + D land in the same lane and register index of two
// different accumulator tiles, so the gate needs no cross-lane traffic at all.
// Weights come as fragment-major images (wn_common.h) read with coalesced 16 B/lane loads
// that every wave on the chip shares through L1/L2.
#include "wn_kernels.h"

template <int R32, int D32, int KS, int MINW>
__global__ __launch_bounds__(256, MINW) void wn_layer_fwd_kernel(WnLayerFwdArgs a) {
  constexpr int R = 32 * R32, D = 32 * D32;
  constexpr int JU = 2 * D32;      // row tiles of u
  constexpr int QR = R / 8;        // k-quads per tap
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= (int64_t)a.B * tiles_per_b) return;     // wave-uniform, no barriers in this kernel
  const int b = (int)(tile / tiles_per_b);
  const int t0 = (int)(tile % tiles_per_b) * 32;
  const int tl = lane & 31, h = lane >> 5;
  const int t = t0 + tl;
  const bool tin = t < a.T;
  const int64_t row = (int64_t)b * a.T + (tin ? t : 0);

  // ---- every activation load of the tile is issued up front (HBM latency), KS taps ----
  //      loads are unconditional from a clamped row; the causal zero padding is applied where the
  //      value is consumed, so that no load is followed by a wait
  f32x4 xq[KS][QR];
  bool xvalid[KS];
#pragma unroll
  for (int tap = 0; tap < KS; ++tap) {
    const int ts = a.xt[tap] ? t : t - (KS - 1 - tap) * a.dilation;
    xvalid[tap] = tin && ts >= 0;
    const float* xrow = (a.xt[tap] ? a.xt[tap] : a.x) + ((int64_t)b * a.T + (xvalid[tap] ? ts : 0)) * R + 4 * h;
#pragma unroll
    for (int q = 0; q < QR; ++q) {
      xq[tap][q] = *reinterpret_cast<const f32x4*>(xrow + 8 * q);
    }
  }

  // ---- u accumulators start at the bias (+ per-utterance conditioning bias); all loads are
  //      issued back to back straight into the accumulator registers ----
  f32x16 u[JU];
#pragma unroll
  for (int j = 0; j < JU; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias_d + 32 * j + 8 * rq + 4 * h);
      u[j][4 * rq + 0] = bv.x; u[j][4 * rq + 1] = bv.y;
      u[j][4 * rq + 2] = bv.z; u[j][4 * rq + 3] = bv.w;
    }
  if (a.cb) {   // wave-uniform
    const float* cbp = a.cb + (int64_t)b * 2 * D + 4 * h;
    f32x4 cv[JU][4];
#pragma unroll
    for (int j = 0; j < JU; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) cv[j][rq] = *reinterpret_cast<const f32x4*>(cbp + 32 * j + 8 * rq);
#pragma unroll
    for (int j = 0; j < JU; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u[j][4 * rq + 0] += cv[j][rq].x; u[j][4 * rq + 1] += cv[j][rq].y;
        u[j][4 * rq + 2] += cv[j][rq].z; u[j][4 * rq + 3] += cv[j][rq].w;
      }
  }

  // ---- dilated causal conv: KS*QR steps, weight fragments streamed PD steps ahead through a
  //      register ring (the compiler otherwise parks every load right in front of its MFMAs) ----
  constexpr int PD = 2;
  constexpr int NS1 = KS * QR;
  const f32x4* frd = reinterpret_cast<const f32x4*>(a.frag_d) + lane;   // step s -> blocks (s*JU + j)
  f32x4 ring[PD + 1][JU];
  wn_static_for<PD>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    if constexpr (st < NS1) {
#pragma unroll
      for (int j = 0; j < JU; ++j) ring[st % (PD + 1)][j] = frd[(st * JU + j) * 64];
    }
  });
  wn_static_for<NS1>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    constexpr int tap = st / QR, q = st % QR;
    if constexpr (st + PD < NS1) {
#pragma unroll
      for (int j = 0; j < JU; ++j) {
        ring[(st + PD) % (PD + 1)][j] = frd[((st + PD) * JU + j) * 64];
      }
    }
    // keep the prefetch where it was issued: the scheduler otherwise sinks it next to its use
    __builtin_amdgcn_sched_barrier(0);
    f32x4 xv = xq[tap][q];
    if (!xvalid[tap]) xv = f32x4{0.f, 0.f, 0.f, 0.f};
    // tile-interleaved order: consecutive MFMAs write different accumulators
#pragma unroll
    for (int j = 0; j < JU; ++j) u[j] = wn_mfma(ring[st % (PD + 1)][j].x, xv.x, u[j]);
#pragma unroll
    for (int j = 0; j < JU; ++j) u[j] = wn_mfma(ring[st % (PD + 1)][j].y, xv.y, u[j]);
#pragma unroll
    for (int j = 0; j < JU; ++j) u[j] = wn_mfma(ring[st % (PD + 1)][j].z, xv.z, u[j]);
#pragma unroll
    for (int j = 0; j < JU; ++j) u[j] = wn_mfma(ring[st % (PD + 1)][j].w, xv.w, u[j]);
  });

  // ---- 1x1 conv weight fragments: start streaming before the (VALU-heavy) gate ----
  constexpr int NS2 = D32 * 4;
  constexpr int PD2 = (R32 <= 2) ? 4 : 2;
  const f32x4* frr = reinterpret_cast<const f32x4*>(a.frag_r) + lane;
  f32x4 ring2[PD2 + 1][R32];
  wn_static_for<PD2>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    if constexpr (st < NS2) {
#pragma unroll
      for (int j = 0; j < R32; ++j) ring2[st % (PD2 + 1)][j] = frr[(st * R32 + j) * 64];
    }
  });

  __builtin_amdgcn_sched_barrier(0);

  // ---- gate: z = tanh(filter) * sigmoid(gate), in place in the filter tiles ----
#pragma unroll
  for (int j = 0; j < D32; ++j) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 av, gv, zv;
      av.x = wn_tanh_fast(u[j][4 * rq + 0]); gv.x = wn_sigmoid_fast(u[j + D32][4 * rq + 0]);
      av.y = wn_tanh_fast(u[j][4 * rq + 1]); gv.y = wn_sigmoid_fast(u[j + D32][4 * rq + 1]);
      av.z = wn_tanh_fast(u[j][4 * rq + 2]); gv.z = wn_sigmoid_fast(u[j + D32][4 * rq + 2]);
      av.w = wn_tanh_fast(u[j][4 * rq + 3]); gv.w = wn_sigmoid_fast(u[j + D32][4 * rq + 3]);
      zv.x = av.x * gv.x; zv.y = av.y * gv.y; zv.z = av.z * gv.z; zv.w = av.w * gv.w;
      u[j][4 * rq + 0] = zv.x; u[j][4 * rq + 1] = zv.y;
      u[j][4 * rq + 2] = zv.z; u[j][4 * rq + 3] = zv.w;
      const int n0 = 32 * j + 8 * rq + 4 * h;
      if (tin) {
        if (a.ag_out) *reinterpret_cast<f32x4*>(a.ag_out + row * D + n0) = gv;
        if (a.z_out) *reinterpret_cast<f32x4*>(a.z_out + row * a.ldz + n0) = zv;
      }
    }
  }

  // ---- 1x1 residual conv: K = D, B operand = z tiles as they stand ----
  f32x16 o[R32];
#pragma unroll
  for (int j = 0; j < R32; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias_r + 32 * j + 8 * rq + 4 * h);
      o[j][4 * rq + 0] = bv.x; o[j][4 * rq + 1] = bv.y;
      o[j][4 * rq + 2] = bv.z; o[j][4 * rq + 3] = bv.w;
    }
  wn_static_for<NS2>([&](auto sc) {
    constexpr int st = decltype(sc)::value;
    constexpr int jz = st / 4, qq = st % 4;
    if constexpr (st + PD2 < NS2) {
#pragma unroll
      for (int j = 0; j < R32; ++j) ring2[(st + PD2) % (PD2 + 1)][j] = frr[((st + PD2) * R32 + j) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < R32; ++j) {
      const f32x4 av = ring2[st % (PD2 + 1)][j];
      o[j] = wn_mfma(av.x, u[jz][4 * qq + 0], o[j]);
      o[j] = wn_mfma(av.y, u[jz][4 * qq + 1], o[j]);
      o[j] = wn_mfma(av.z, u[jz][4 * qq + 2], o[j]);
      o[j] = wn_mfma(av.w, u[jz][4 * qq + 3], o[j]);
    }
  });

  if (!tin) return;
#pragma unroll
  for (int j = 0; j < R32; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int n0 = 32 * j + 8 * rq + 4 * h;
      f32x4 ov;
      ov.x = o[j][4 * rq + 0]; ov.y = o[j][4 * rq + 1];
      ov.z = o[j][4 * rq + 2]; ov.w = o[j][4 * rq + 3];
      if (a.o_out) *reinterpret_cast<f32x4*>(a.o_out + row * R + n0) = ov;
      if (a.residual) {
        const f32x4 xr = a.res ? *reinterpret_cast<const f32x4*>(a.res + row * R + n0) : xq[KS - 1][j * 4 + rq];
        ov.x += xr.x; ov.y += xr.y; ov.z += xr.z; ov.w += xr.w;
      }
      *reinterpret_cast<f32x4*>(a.x_out + row * R + n0) = ov;
    }
}

int wn_layer_fwd_supported(int R, int D, int KS) {
  if (KS != 2 && KS != 3) return 0;
  if (R == 32 && D == 32) return 1;
  if (R == 64 && D == 64) return 1;
  if (R == 128 && D == 128) return KS == 2;
  return 0;
}

#define WN_LAUNCH_LAYER(R32, D32, KS, MINW) \
  hipLaunchKernelGGL((wn_layer_fwd_kernel<R32, D32, KS, MINW>), dim3((unsigned)gx), dim3(256), 0, s, a)

int wn_launch_layer_fwd(const WnLayerFwdArgs& a, hipStream_t s) {
  if (!wn_layer_fwd_supported(a.R, a.D, a.KS)) {
    wn_set_error("layer_fwd: unsupported shape R=%d D=%d KS=%d", a.R, a.D, a.KS);
    return WN_E_UNSUPPORTED;
  }
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  const int64_t gx = (tiles + 3) / 4;
  if (gx <= 0) return WN_OK;
  if (a.R == 32) {
    if (a.KS == 2) WN_LAUNCH_LAYER(1, 1, 2, 2);
    else WN_LAUNCH_LAYER(1, 1, 3, 2);
  } else if (a.R == 64) {
    if (a.KS == 2) WN_LAUNCH_LAYER(2, 2, 2, 2);
    else WN_LAUNCH_LAYER(2, 2, 3, 2);
  } else {
    WN_LAUNCH_LAYER(4, 4, 2, 1);
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
