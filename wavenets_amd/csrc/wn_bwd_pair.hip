// Backward-data chain, two products per launch (gfx950, split-precision MFMA):
//   g_x(b+1)[t] = W_0(b+1) g_u(b+1)[t + d] + W_1(b+1) g_u(b+1)[t] + g_x(b+2)[t]        (dilated conv of block b+1, reversed;
//                                                                                        + the residual path)
//   g_u(b)[t]   = gate'( W_r(b) g_x(b+1)[t] + V(b) dL/da[t] )                            (1x1 conv + folded skip path of block b)
// (src/layers.py:199-223 reversed; V(b) = W_s(b) W_f0, see wn_skip_fold in wn_elem.hip).  g_x(b+1) is the output gradient
// of block b at the SAME rows, so the tile a wave has just produced stays in its registers -- a 32x32 accumulator tile is,
// unchanged, the B operand of the next contraction over its channel index (wn_layer16.hip uses the same fact for z) --
// instead of going to HBM and back between two launches.  One launch per block boundary instead of two: every launch of
// the chain pays ~17 us of ramp (weight images -> LDS, first HBM latency) and drain on top of its bytes.
//
// Structure = the resident rows GEMM of wn_gemm16.hip twice: persistent workgroups of 8 waves, one 32-row tile per wave
// at a time, both weight images resident in LDS (A[64][256] of the reversed conv: 64 KiB; [W_r | V]: 48 KiB), a rolling
// register ring of 4 k-steps of activations that runs through both products and across tile boundaries (24 "memory"
// k-steps per tile: 8 of g_u[t + d], 8 of g_u[t], 8 of dL/da[t]; the 4 k-steps over g_x come from registers), epilogue
// operands requested ahead, outputs through a wave-private LDS stage of 32 columns (the images leave 36 KiB for stages) as
// 128-byte row segments.  Operand scaling: the first product uses the running max-abs of g_u(b+1) like the unfused
// kernel (bit-identical g_x); the second one cannot know the max-abs of the tensor it is producing, so every tile is
// scaled by the exact power of two of max(its own max |g_x|, running max-abs of dL/da).
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 bp_h8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ f32x16 bp_mfma(bp_h8 a, bp_h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void bp_split8(const f32x4& q0, const f32x4& q1, float s, bp_h8& hi, bp_h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)(v[e] * s);
    hi[e] = h;
    lo[e] = (_Float16)__builtin_fmaf(v[e], s, -(float)h);
  }
}

__device__ __forceinline__ f32x4 bp_ldg4(const float* p) { return *(const __attribute__((address_space(1))) f32x4*)(p); }

__device__ __forceinline__ void bp_pow2_scale(float m, float& sc, float& inv) {
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

constexpr int BP_NK1 = 16;                       // k-steps of the reversed conv (KS * 2D / 16)
constexpr int BP_NKR = 4;                        // k-steps over g_x (R / 16), operands from registers
constexpr int BP_NKF = 8;                        // k-steps over dL/da (F0 / 16)
constexpr int BP_NKM = BP_NK1 + BP_NKF;          // memory k-steps per tile
constexpr int BP_W1 = BP_NK1 * 2 * 2048, BP_W2 = (BP_NKR + BP_NKF) * 2 * 2048;
constexpr int BP_PITCH = 36;
constexpr int BP_STAGE = 32 * BP_PITCH * 4;      // 4608
constexpr int BP_LDS = BP_W1 + BP_W2 + 8 * BP_STAGE;   // 151552
constexpr int BP_PF = 4;

// one 32 x 32 tile (D layout) -> LDS stage -> 128-byte row segments
__device__ __forceinline__ void bp_store32(const f32x16& v, float* stage, float* dst, int64_t ld, int rows_valid, int lane) {
  const int tl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 o;
    o.x = v[4 * rq + 0]; o.y = v[4 * rq + 1]; o.z = v[4 * rq + 2]; o.w = v[4 * rq + 3];
    *reinterpret_cast<f32x4*>(stage + tl * BP_PITCH + 8 * rq + 4 * h) = o;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const int col = (lane & 7) * 4;
    const f32x4 o = *reinterpret_cast<const f32x4*>(stage + row * BP_PITCH + col);
    if (row < rows_valid) *reinterpret_cast<f32x4*>(dst + (int64_t)row * ld + col) = o;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

}  // namespace

__global__ __launch_bounds__(512, 2) void wn_bwd_pair_kernel(WnBwdPairArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BP_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  float* stage = reinterpret_cast<float*>(smem + BP_W1 + BP_W2 + wave * BP_STAGE);
  {
    // both images in ONE round trip (wn_images_to_lds: every load in flight before the first LDS store)
    wn_images_to_lds<512, BP_W1 / 16, BP_W2 / 16>(a.wx16, smem, BP_W1 / 16, a.wu16, smem + BP_W1, BP_W2 / 16, tid);
  }
  __syncthreads();
  const bp_h8* w1 = reinterpret_cast<const bp_h8*>(smem) + lane;
  const bp_h8* w2 = reinterpret_cast<const bp_h8*>(smem + BP_W1) + lane;

  float sc1, inv1;
  bp_pow2_scale(a.am_gu_in ? *a.am_gu_in : 0.f, sc1, inv1);
  const float gfmax = a.am_gf ? *a.am_gf : 0.f;

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
  // memory k-step m of a tile: 0..7 g_u(b+1)[t + d], 8..15 g_u(b+1)[t], 16..23 dL/da[t]  (straight-line, unconditional:
  // masked rows read a clamped row and are zeroed where they are consumed)
  auto load_x = [&](const TileCtx& c, int m, f32x4& q0, f32x4& q1, bool& okout) {
    const int seg = m >> 3;                                        // wave-uniform
    const int kk = m & 7;
    const int ts = c.t + (seg == 0 ? a.dil : 0);
    const bool ok = c.t < a.T && ts < a.T;
    const float* base = seg == 2 ? a.gf : a.gu_in;
    const float* src = base + ((int64_t)c.b * a.T + (ok ? ts : 0)) * 128 + 4 * h + 16 * kk;
    q0 = bp_ldg4(src);
    q1 = bp_ldg4(src + 8);
    okout = ok;
  };

  float wmax_x = 0.f, wmax_u = 0.f;
  const WnTileWalk walk = wn_tile_walk(ntiles, 8, wave);          // each XCD a contiguous eighth of the tiles (wn_common.h)
  const int64_t tstride = walk.stride, tend = walk.end;
  int64_t tile = walk.first;
  TileCtx cur, nxt;
  f32x4 xr[BP_PF][2];
  bool okr[BP_PF];
  if (tile < tend) {
    make_ctx(tile, cur);
#pragma unroll
    for (int k = 0; k < BP_PF; ++k) load_x(cur, k, xr[k][0], xr[k][1], okr[k]);
  }
  for (; tile < tend; tile += tstride) {
    const bool has_next = tile + tstride < tend;                   // wave-uniform
    if (has_next) make_ctx(tile + tstride, nxt);
    const bool tin = cur.t < a.T;
    const int64_t row = cur.row0 + (tin ? tl : 0);                  // rows past the end read the tile's first row (unused)

    // ---- residual operand of the first product, requested before its contraction ----
    f32x4 addc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) addc[j][rq] = bp_ldg4(a.gx_res + row * 64 + 32 * j + 8 * rq + 4 * h);

    // ---- g_x(b+1) = reversed dilated conv of g_u(b+1) ----
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int m0 = 0; m0 < BP_NK1; m0 += BP_PF) {
      wn_static_for<BP_PF>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        bp_h8 bh, bl;
        bp_split8(xr[k][0], xr[k][1], okr[k] ? sc1 : 0.f, bh, bl);
        __builtin_amdgcn_sched_barrier(0);
        load_x(cur, m0 + k + BP_PF, xr[k][0], xr[k][1], okr[k]);    // m0 + k + 4 <= 19 < 24: always this tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bp_h8 ah = w1[(((m0 + k) * 2 + j) * 2 + 0) * 64];
          const bp_h8 al = w1[(((m0 + k) * 2 + j) * 2 + 1) * 64];
          acc[j] = bp_mfma(al, bh, acc[j]);
          acc[j] = bp_mfma(ah, bl, acc[j]);
          acc[j] = bp_mfma(ah, bh, acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // epilogue 1: + g_x(b+2) (residual path); rows past the utterance are exact zeros (they feed the next product)
    f32x16 gx[2];
    float tmax = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
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
    // ---- gate-derivative operands of block b, requested before the second contraction ----
    f32x4 sg[2][4], zz[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        sg[j][rq] = bp_ldg4(a.ag + row * 64 + 32 * j + 8 * rq + 4 * h);
        zz[j][rq] = bp_ldg4(a.z + row * a.ldz + 32 * j + 8 * rq + 4 * h);
      }
    if (cur.rows_valid > 0) {
      bp_store32(gx[0], stage, a.gx_out + cur.row0 * 64, 64, cur.rows_valid, lane);
      bp_store32(gx[1], stage, a.gx_out + cur.row0 * 64 + 32, 64, cur.rows_valid, lane);
    }
    wmax_x = fmaxf(wmax_x, tmax);
    // per-tile power-of-two scale of the second product's B operands
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    float sc2, inv2;
    bp_pow2_scale(fmaxf(tmax, gfmax), sc2, inv2);

    // ---- g_z = W_r g_x (registers: a D tile IS the B operand over its channel index) + V dL/da (memory) ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    wn_static_for<BP_NKR>([&](auto sc) {
      constexpr int ks = decltype(sc)::value;
      constexpr int jz = ks / 2, r0 = 8 * (ks % 2);
      f32x4 q0, q1;
      q0.x = gx[jz][r0 + 0]; q0.y = gx[jz][r0 + 1]; q0.z = gx[jz][r0 + 2]; q0.w = gx[jz][r0 + 3];
      q1.x = gx[jz][r0 + 4]; q1.y = gx[jz][r0 + 5]; q1.z = gx[jz][r0 + 6]; q1.w = gx[jz][r0 + 7];
      bp_h8 bh, bl;
      bp_split8(q0, q1, sc2, bh, bl);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bp_h8 ah = w2[((ks * 2 + j) * 2 + 0) * 64];
        const bp_h8 al = w2[((ks * 2 + j) * 2 + 1) * 64];
        acc[j] = bp_mfma(al, bh, acc[j]);
        acc[j] = bp_mfma(ah, bl, acc[j]);
        acc[j] = bp_mfma(ah, bh, acc[j]);
      }
    });
    for (int m0 = BP_NK1; m0 < BP_NKM; m0 += BP_PF) {
      wn_static_for<BP_PF>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        bp_h8 bh, bl;
        bp_split8(xr[k][0], xr[k][1], okr[k] ? sc2 : 0.f, bh, bl);
        __builtin_amdgcn_sched_barrier(0);
        const int mn = m0 + k + BP_PF;
        if (mn < BP_NKM) load_x(cur, mn, xr[k][0], xr[k][1], okr[k]);
        else if (has_next) load_x(nxt, mn - BP_NKM, xr[k][0], xr[k][1], okr[k]);
        const int ks2 = BP_NKR + (m0 + k - BP_NK1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bp_h8 ah = w2[((ks2 * 2 + j) * 2 + 0) * 64];
          const bp_h8 al = w2[((ks2 * 2 + j) * 2 + 1) * 64];
          acc[j] = bp_mfma(al, bh, acc[j]);
          acc[j] = bp_mfma(ah, bl, acc[j]);
          acc[j] = bp_mfma(ah, bh, acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // epilogue 2: gate derivative (filter half -> columns 0..63, gate half -> columns 64..127 of g_u)
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 o;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float gv[4] = {sg[j][rq].x, sg[j][rq].y, sg[j][rq].z, sg[j][rq].w};
          const float zv[4] = {zz[j][rq].x, zz[j][rq].y, zz[j][rq].z, zz[j][rq].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dz = acc[j][4 * rq + e] * inv2;
            const float v = tin ? (part == 0 ? wn_gate_bwd_f(dz, gv[e], zv[e]) : wn_gate_bwd_g(dz, gv[e], zv[e])) : 0.f;
            o[4 * rq + e] = v;
            wmax_u = fmaxf(wmax_u, fabsf(v));
          }
        }
        if (cur.rows_valid > 0) bp_store32(o, stage, a.gu_out + cur.row0 * 128 + 64 * part + 32 * j, 128, cur.rows_valid, lane);
      }
    if (has_next) cur = nxt;
  }
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

int wn_bwd_pair_supported(int R, int D, int KS, int F0) { return (R == 64 && D == 64 && KS == 2 && F0 == 128) ? 1 : 0; }

int wn_launch_bwd_pair(const WnBwdPairArgs& a, hipStream_t s) {
  const int64_t tiles = (int64_t)a.B * ((a.T + 31) / 32);
  if (tiles <= 0) return WN_OK;
  int64_t gx = (tiles + 7) / 8;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(wn_bwd_pair_kernel, dim3((unsigned)gx), dim3(512), 0, s, a);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
