// Fused WaveNet residual block forward, split-precision MFMA variant (gfx950).
// Reference: WaveNetLayer.call, src/layers.py:178-224 (depth-1 dilated stack).
//
// Same chain and register orientation as wn_layer.hip (time on lanes; accumulator tiles feed the
// next contraction unchanged), but every fp32 product a*b is evaluated as
//     a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,   a = a_hi + a_lo  (fp16 hi / lo, fp32 accumulate)
// on v_mfma_f32_32x32x16_f16: 3 MFMAs of 16 k each instead of 8 fp32 MFMAs of 2 k -> 5.3x fewer
// matrix-pipe cycles at |error| <= ~6e-7 on O(1) results (tools/f16_split_probe.hip; the MFMA
// keeps fp16 subnormals, so the lo parts never flush).  That moves the block from the fp32-MFMA
// bound to the HBM bound the roofline contract asks for, so the rest of the kernel is built as a
// streaming kernel:
//   * persistent workgroups (8 waves, one per CU), weight images (hi|lo) resident in LDS
//   * every output tile (x_out, z, tanh, sigmoid) is transposed through a wave-private LDS
//     stage and leaves as full 256-byte row segments (1 KiB per wave store instruction)
//   * no barrier inside the tile loop: waves only share the read-only weights.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 wn_mfma16(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// split 8 fp32 values (two float4 quads) into fp16 hi / lo fragments
__device__ __forceinline__ void wn_split8(const f32x4& q0, const f32x4& q1, h8& hi, h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)v[e];
    hi[e] = h;
    lo[e] = (_Float16)(v[e] - (float)h);
  }
}

template <int R32, int D32, int KS>
struct WnL16 {
  static constexpr int R = 32 * R32, D = 32 * D32, JU = 2 * D32;
  static constexpr int KS1 = KS * R / 16;        // k-steps of the dilated conv
  static constexpr int KS2 = D / 16;             // k-steps of the 1x1 conv
  static constexpr int WD_BYTES = KS1 * JU * 2048;
  static constexpr int WR_BYTES = KS2 * R32 * 2048;
  static constexpr int CMAX = (R > D ? R : D);
  static constexpr int PITCH = CMAX + 4;         // floats; +4 keeps b128 accesses conflict-free
  static constexpr int STAGE_BYTES = 32 * PITCH * 4;
  static constexpr int WAVES = 8;
  static constexpr int BIAS_BYTES = (2 * D + R) * 4;          // b_d | b_r, staged once per workgroup
  static constexpr int LDS_BYTES = WD_BYTES + WR_BYTES + WAVES * STAGE_BYTES + BIAS_BYTES;
};

// tile (32 rows x C floats) held as D-layout registers -> LDS stage -> coalesced rows in HBM
template <int C32, int PITCH>
__device__ __forceinline__ void wn_store_tile(const f32x16 (&v)[C32], float* stage, float* dst, int64_t ld,
                                              int rows_valid, int lane) {
  const int tl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < C32; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 o;
      o.x = v[j][4 * rq + 0]; o.y = v[j][4 * rq + 1]; o.z = v[j][4 * rq + 2]; o.w = v[j][4 * rq + 3];
      *reinterpret_cast<f32x4*>(stage + tl * PITCH + 32 * j + 8 * rq + 4 * h) = o;
    }
  // LDS instructions of one wave execute in order, so the reads below see every lane's writes;
  // the asm statements only stop the compiler from moving LDS accesses across the two phases
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int LPR = C32 * 8;                 // lanes (16-byte pieces) per row
  constexpr int RPI = 64 / LPR;                // rows per store instruction
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    const int row = i * RPI + lane / LPR;
    const int col = (lane % LPR) * 4;
    const f32x4 o = *reinterpret_cast<const f32x4*>(stage + row * PITCH + col);
    if (row < rows_valid) *reinterpret_cast<f32x4*>(dst + (int64_t)row * ld + col) = o;
  }
  asm volatile("" ::: "memory");
}

// FAST = the plain training / inference call (taps from x with the block's dilation, no conditioning bias,
// residual from the newest tap, z written, no separate pre-residual output): the optional paths and
// their registers are compiled out.  FAST == 2 additionally saves the sigmoid (training); FAST == 3 is FAST == 2 with the
// per-utterance conditioning bias kept (training passes of globally conditioned networks).
template <int R32, int D32, int KS, int FAST>
__global__ __launch_bounds__(512, 2) void wn_layer_fwd_f16_kernel(WnLayerFwdArgs a_in) {
  WnLayerFwdArgs a = a_in;
  if constexpr (FAST > 0) {
    a.xt[0] = a.xt[1] = a.xt[2] = nullptr; a.res = nullptr; a.o_out = nullptr; a.residual = 1;
    if constexpr (FAST != 3) a.cb = nullptr;
    if constexpr (FAST == 1) a.ag_out = nullptr;
  }
  using G = WnL16<R32, D32, KS>;
  constexpr int R = G::R, D = G::D, JU = G::JU, KS1 = G::KS1, KS2 = G::KS2, PITCH = G::PITCH;
  constexpr int QR = R / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;

  // ---- weights: global (fp16 hi|lo images) -> LDS, once per workgroup ----
  {
    // the biases too: per tile they would be 6 * (R + D) / 32 global loads whose results are needed at once.  Requested
    // first, so that they travel with the image loads (one round trip for everything, see wn_images_to_lds)
    float* sb = reinterpret_cast<float*>(smem + G::WD_BYTES + G::WR_BYTES + G::WAVES * G::STAGE_BYTES);
    static_assert(2 * D + R <= 512, "one bias element per thread");
    const float bval = tid < 2 * D ? a.bias_d[tid] : (tid < 2 * D + R ? a.bias_r[tid - 2 * D] : 0.f);
    wn_images_to_lds<512, G::WD_BYTES / 16, G::WR_BYTES / 16>(a.frag_d, smem, G::WD_BYTES / 16, a.frag_r, smem + G::WD_BYTES,
                                                               G::WR_BYTES / 16, tid);
    if (tid < 2 * D + R) sb[tid] = bval;
  }
  __syncthreads();
  const float* lbias_d = reinterpret_cast<const float*>(smem + G::WD_BYTES + G::WR_BYTES + G::WAVES * G::STAGE_BYTES);
  const float* lbias_r = lbias_d + 2 * D;
  const h8* wd = reinterpret_cast<const h8*>(smem) + lane;                 // block b -> wd[b * 64]
  const h8* wr = reinterpret_cast<const h8*>(smem + G::WD_BYTES) + lane;
  float* stage = reinterpret_cast<float*>(smem + G::WD_BYTES + G::WR_BYTES + wave * G::STAGE_BYTES);

  const int tiles_per_b = (a.T + 31) >> 5;
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  float wmax = 0.f;                                      // forward range guard: max |x_out| of this wave's tiles
  const WnTileWalk walk = wn_tile_walk(ntiles, G::WAVES, wave);
  for (int64_t tile = walk.first; tile < walk.end; tile += walk.stride) {
    const int b = (int)(tile / tiles_per_b);
    const int t0 = (int)(tile % tiles_per_b) * 32;
    const int t = t0 + tl;
    const bool tin = t < a.T;
    const int rows_valid = min(32, a.T - t0);
    const int64_t row0 = (int64_t)b * a.T + t0;

    // ---- activation loads of all taps, issued up front (unconditional, clamped row) ----
    f32x4 xq[KS][QR];
    bool xvalid[KS];
#pragma unroll
    for (int tap = 0; tap < KS; ++tap) {
      const int ts = a.xt[tap] ? t : t - (KS - 1 - tap) * a.dilation;
      xvalid[tap] = tin && ts >= 0;
      const float* xrow = (a.xt[tap] ? a.xt[tap] : a.x) + ((int64_t)b * a.T + (xvalid[tap] ? ts : 0)) * R + 4 * h;
#pragma unroll
      for (int q = 0; q < QR; ++q) xq[tap][q] = *reinterpret_cast<const f32x4*>(xrow + 8 * q);
    }
    // the per-utterance conditioning bias: ONE coalesced request per lane (2 D floats a row) in the same round trip as the
    // activations, spread to the accumulator layout through the wave's (still idle) output stage.  (As 16 D-layout loads
    // behind the pin it cost a second, exposed round trip per tile: 38.8 against 34.5 us a launch with global conditioning.)
    f32x4 cbv = {0.f, 0.f, 0.f, 0.f};
    if (a.cb && 4 * lane < 2 * D) cbv = *reinterpret_cast<const f32x4*>(a.cb + (int64_t)b * 2 * D + 4 * lane);
    // Pin the requests HERE.  Without it hipcc's scheduler sinks most of them into the conv below, each right in front of its
    // first use with s_waitcnt vmcnt(0) behind it (to shorten the live ranges of the 2 * R / 8 quads): the conv then walks
    // six exposed global round trips per tile (found with s_memtime stamps: 8.1 k clocks for 96 MFMAs).
    __builtin_amdgcn_sched_barrier(0);
    // ---- accumulators start at the bias (+ per-utterance conditioning bias) ----
    f32x16 u[JU];
#pragma unroll
    for (int j = 0; j < JU; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias_d + 32 * j + 8 * rq + 4 * h);
        u[j][4 * rq + 0] = bv.x; u[j][4 * rq + 1] = bv.y; u[j][4 * rq + 2] = bv.z; u[j][4 * rq + 3] = bv.w;
      }
    if (a.cb) {   // wave-uniform
      f32x4* cbl = reinterpret_cast<f32x4*>(stage);
      if (4 * lane < 2 * D) cbl[lane] = cbv;
#pragma unroll
      for (int j = 0; j < JU; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 cv = cbl[8 * j + 2 * rq + h];      // floats 32 j + 8 rq + 4 h ..
          u[j][4 * rq + 0] += cv.x; u[j][4 * rq + 1] += cv.y; u[j][4 * rq + 2] += cv.z; u[j][4 * rq + 3] += cv.w;
        }
    }

    // ---- dilated causal conv: KS1 k-steps of 16, 3 MFMAs per (k-step, tile) ----
    // The weight fragments come out of LDS through a register ring two (k-step, tile) blocks ahead, pinned with
    // sched_barrier: left to itself hipcc reads each fragment right in front of its products (ds_read, s_waitcnt lgkmcnt(0),
    // MFMA: ~130 clocks of LDS latency in front of every 96 clocks of matrix work; s_memtime stamps: 8.1 k clocks for a
    // tile's 96 products, and a wave's phases are latency chains that a partner wave does not fill).
    {
      constexpr int NB = KS1 * JU;                       // (k-step, tile) blocks in consumption order
      h8 fr[3][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { fr[i][0] = wd[(i * 2 + 0) * 64]; fr[i][1] = wd[(i * 2 + 1) * 64]; }
      h8 bh, bl;
      wn_static_for<NB>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;
        constexpr int ks = blk / JU, j = blk % JU;
        if constexpr (j == 0) {
          constexpr int tap = ks / (R / 16), kk = ks % (R / 16);
          f32x4 q0 = xq[tap][2 * kk], q1 = xq[tap][2 * kk + 1];
          if (!xvalid[tap]) { q0 = f32x4{0.f, 0.f, 0.f, 0.f}; q1 = q0; }
          wn_split8(q0, q1, bh, bl);
        }
        if constexpr (blk + 2 < NB) {
          fr[(blk + 2) % 3][0] = wd[((blk + 2) * 2 + 0) * 64];
          fr[(blk + 2) % 3][1] = wd[((blk + 2) * 2 + 1) * 64];
        }
        u[j] = wn_mfma16(fr[blk % 3][1], bh, u[j]);
        u[j] = wn_mfma16(fr[blk % 3][0], bl, u[j]);
        u[j] = wn_mfma16(fr[blk % 3][0], bh, u[j]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- gate; a / g tiles reuse the accumulator registers ----
    //      u[j] (filter) -> z, u[j + D32] (gate) -> sigmoid; tanh kept in a separate tile set only
    //      while it is being written out
    if (a.ag_out) {
      // the sigmoid tile is saved for backward (tanh is recovered there as z / sigmoid)
#pragma unroll
      for (int j = 0; j < D32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          u[j + D32][r] = wn_sigmoid_fast(u[j + D32][r]);
          u[j][r] = wn_tanh_fast(u[j][r]) * u[j + D32][r];
        }
      if (rows_valid > 0) {
        f32x16 gv[D32];
#pragma unroll
        for (int j = 0; j < D32; ++j) gv[j] = u[j + D32];
        wn_store_tile<D32, PITCH>(gv, stage, a.ag_out + row0 * D, D, rows_valid, lane);
      }
    } else {
#pragma unroll
      for (int j = 0; j < D32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) u[j][r] = wn_tanh_fast(u[j][r]) * wn_sigmoid_fast(u[j + D32][r]);
    }
    if (a.z_out && rows_valid > 0) {
      f32x16 zv[D32];
#pragma unroll
      for (int j = 0; j < D32; ++j) zv[j] = u[j];
      wn_store_tile<D32, PITCH>(zv, stage, a.z_out + row0 * a.ldz, a.ldz, rows_valid, lane);
    }

    // ---- 1x1 residual conv: B operand = z tiles as they stand (rows 8s..8s+7 of a tile = k-step) ----
    f32x16 o[R32];
#pragma unroll
    for (int j = 0; j < R32; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias_r + 32 * j + 8 * rq + 4 * h);
        o[j][4 * rq + 0] = bv.x; o[j][4 * rq + 1] = bv.y; o[j][4 * rq + 2] = bv.z; o[j][4 * rq + 3] = bv.w;
      }
    {
      constexpr int NB = KS2 * R32;                      // the same fragment ring as in the conv
      h8 fr[3][2];
#pragma unroll
      for (int i = 0; i < 2 && i < NB; ++i) { fr[i][0] = wr[(i * 2 + 0) * 64]; fr[i][1] = wr[(i * 2 + 1) * 64]; }
      h8 bh, bl;
      wn_static_for<NB>([&](auto bc) {
        constexpr int blk = decltype(bc)::value;
        constexpr int ks = blk / R32, j = blk % R32;
        if constexpr (j == 0) {
          constexpr int jz = ks / 2, r0 = 8 * (ks % 2);
          f32x4 q0, q1;
          q0.x = u[jz][r0 + 0]; q0.y = u[jz][r0 + 1]; q0.z = u[jz][r0 + 2]; q0.w = u[jz][r0 + 3];
          q1.x = u[jz][r0 + 4]; q1.y = u[jz][r0 + 5]; q1.z = u[jz][r0 + 6]; q1.w = u[jz][r0 + 7];
          wn_split8(q0, q1, bh, bl);
        }
        if constexpr (blk + 2 < NB) {
          fr[(blk + 2) % 3][0] = wr[((blk + 2) * 2 + 0) * 64];
          fr[(blk + 2) % 3][1] = wr[((blk + 2) * 2 + 1) * 64];
        }
        o[j] = wn_mfma16(fr[blk % 3][1], bh, o[j]);
        o[j] = wn_mfma16(fr[blk % 3][0], bl, o[j]);
        o[j] = wn_mfma16(fr[blk % 3][0], bh, o[j]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    if (rows_valid > 0) {
      if (a.o_out) wn_store_tile<R32, PITCH>(o, stage, a.o_out + row0 * R, R, rows_valid, lane);
      if (a.residual) {
#pragma unroll
        for (int j = 0; j < R32; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            f32x4 xr;
            if (a.res) xr = tin ? *reinterpret_cast<const f32x4*>(a.res + (row0 + tl) * R + 32 * j + 8 * rq + 4 * h)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
            else xr = xq[KS - 1][j * 4 + rq];
            o[j][4 * rq + 0] += xr.x; o[j][4 * rq + 1] += xr.y; o[j][4 * rq + 2] += xr.z; o[j][4 * rq + 3] += xr.w;
          }
      }
      if (tin) {
#pragma unroll
        for (int j = 0; j < R32; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) wmax = fmaxf(wmax, fabsf(o[j][r]));
      }
      wn_store_tile<R32, PITCH>(o, stage, a.x_out + row0 * R, R, rows_valid, lane);
    }
  }
  if (a.absmax_out) {
    if (!(wmax < 3.0e38f)) wmax = 3.0e38f;               // inf / NaN: beyond any limit
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
    if (lane == 0) wn_absmax_publish_any(a.absmax_out, wmax);
  }
}

size_t wn_frag16_floats(int I, int KK) { return (size_t)((KK + 15) / 16) * ((I + 31) / 32) * 512; }

int wn_layer_fwd_f16_supported(int R, int D, int KS) {
  if (R == 32 && D == 32) return KS == 2 || KS == 3;
  if (R == 64 && D == 64) return KS == 2;    // LDS: 64 KiB W_d + 16 KiB W_r + 68 KiB stages
  return 0;
}

int wn_launch_layer_fwd_f16(const WnLayerFwdArgs& a, hipStream_t s) {
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  if (tiles <= 0) return WN_OK;
  int64_t gx = (tiles + 7) / 8;
  if (gx > 256) gx = 256;                    // one persistent workgroup per CU
  const bool plain = !a.xt[0] && !a.xt[1] && !a.xt[2] && !a.res && !a.o_out && a.residual && a.z_out;
  const int fast = (plain && a.ag_out) ? (a.cb ? 3 : 2) : 0;       // (the inference form, FAST == 1, spills: generic kernel)
#define WN_L16_LAUNCH(R32_, D32_, KS_)                                                                                      \
  do {                                                                                                                      \
    if (fast == 2) hipLaunchKernelGGL((wn_layer_fwd_f16_kernel<R32_, D32_, KS_, 2>), dim3((unsigned)gx), dim3(512), 0, s, a);      \
    else if (fast == 3) hipLaunchKernelGGL((wn_layer_fwd_f16_kernel<R32_, D32_, KS_, 3>), dim3((unsigned)gx), dim3(512), 0, s, a); \
    else hipLaunchKernelGGL((wn_layer_fwd_f16_kernel<R32_, D32_, KS_, 0>), dim3((unsigned)gx), dim3(512), 0, s, a);                \
  } while (0)
  if (a.R == 32 && a.KS == 2) WN_L16_LAUNCH(1, 1, 2);
  else if (a.R == 32 && a.KS == 3) WN_L16_LAUNCH(1, 1, 3);
  else if (a.R == 64 && a.KS == 2) WN_L16_LAUNCH(2, 2, 2);
  else { wn_set_error("layer_fwd_f16: unsupported shape R=%d D=%d KS=%d", a.R, a.D, a.KS); return WN_E_UNSUPPORTED; }
#undef WN_L16_LAUNCH
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
