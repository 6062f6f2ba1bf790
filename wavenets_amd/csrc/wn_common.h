// Common device-side helpers for the gfx950 WaveNet kernels.
//
// Orientation used by every contraction kernel ("time on lanes"):
//   D[i = channel][j = time] += A[i][k] * B[k][j]     v_mfma_f32_32x32x2_f32 (exact fp32)
//   A operand  (weights)      lane l holds A[i = l&31][k = l>>5]
//   B operand  (activations)  lane l holds B[k = l>>5][j = l&31]
//   C/D tile   lane l, reg r  holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// A wave therefore owns 32 consecutive time steps (one per lane pair) and ALL output
// channels of them; a D tile's registers are, unchanged, the B operand of the next
// contraction over its channel index (k order 8q + 4h + e, see wn_frag_index), so the
// dilated conv -> gate -> 1x1 chain never leaves registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wn_hip.h"   // WN_OK / WN_E_* / WN_ACT_* / WN_HEAD_* constants

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum WnEpilogue : int {
  WN_EPI_PLAIN = 0,      // y = act(acc + bias + rowbias + addc)
  WN_EPI_DACT = 1,       // y = (acc + addc) * act'(saved y)        (backward data)
  WN_EPI_GATE_BWD = 2,   // acc is dL/dz; y[:, :D] = dL/du_f, y[:, D:] = dL/du_g from saved (a, g)
  WN_EPI_GATE_FWD = 3,   // acc is u = [filter | gate] (column blocks of 64 + 64): y = tanh * sigmoid, y2 = sigmoid
};

__device__ __forceinline__ f32x16 wn_mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of a 32x32 D tile held by register r of a lane in half h = lane >> 5
__device__ __forceinline__ int wn_drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float wn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp
// each): sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 2 sigmoid(2x) - 1.  Absolute error
// <= ~2.5e-7, i.e. 400x below the 1e-4 per-activation budget, at ~10 VALU instructions per
// (filter, gate) pair instead of ~56 for the libm forms.  Saturates correctly (exp2 -> 0 / inf).
__device__ __forceinline__ float wn_sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// The fma is explicit on purpose: with -ffp-contract=fast the compiler may otherwise turn
// (1 - 2r) * s into fma(-2r, s, s) in one kernel and into fma(-2, r, 1) * s in another, and the
// generation kernels must reproduce the training-forward kernels bit for bit.
__device__ __forceinline__ float wn_tanh_fast(float x) {
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)), 1.0f);
}

__device__ __forceinline__ float wn_act(float x, int act) {
  switch (act) {
    case WN_ACT_RELU: return x > 0.f ? x : 0.f;
    case WN_ACT_LEAKY_RELU: return x >= 0.f ? x : 0.2f * x;
    case WN_ACT_TANH: return tanhf(x);
    case WN_ACT_SIGMOID: return wn_sigmoid(x);
    case WN_ACT_ELU: return x > 0.f ? x : expm1f(x);
    default: return x;
  }
}

// derivative of the activation expressed through its OUTPUT y (all supported activations
// are invertible enough for this: sign(y) == sign(x) for relu / leaky / elu)
__device__ __forceinline__ float wn_dact_from_y(float y, int act) {
  switch (act) {
    case WN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case WN_ACT_LEAKY_RELU: return y >= 0.f ? 1.f : 0.2f;
    case WN_ACT_TANH: return 1.f - y * y;
    case WN_ACT_SIGMOID: return y * (1.f - y);
    case WN_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    default: return 1.f;
  }
}

// Fragment-major weight image.  For a matrix A[I][KK] (I output rows, KK contraction) the
// image is a sequence of 1 KiB blocks, one per (k-quad q = kk/8, row tile j = i/32):
//   img[((q * JT + j) * 64 + lane) * 4 + e] = A[32 j + (lane & 31)][8 q + 4 (lane >> 5) + e]
// so that one coalesced 16-byte-per-lane load gives a lane its A operand for 4 MFMA steps.
__host__ __device__ __forceinline__ size_t wn_frag_floats(int I, int KK) {
  return (size_t)((KK + 7) / 8) * ((I + 31) / 32) * 256;
}

// compile-time loop: the body receives std::integral_constant<int, I>, so register arrays
// indexed by it can never be demoted to private memory by a failed unroll
#ifdef __cplusplus
#include <type_traits>
template <int N, class F>
__device__ __forceinline__ void wn_static_for(F&& f) {
  if constexpr (N > 0) {
    wn_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
#endif

#define WN_HIP_CHECK(expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) {                                  \
      wn_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return WN_E_HIP;                                       \
    }                                                        \
  } while (0)

// gate derivative from the saved sigmoid g and z = tanh * g:  a = z / g (g > 0; when g underflowed both
// derivatives vanish anyway);  d/du_f = dz * g * (1 - a^2),  d/du_g = dz * a * g * (1 - g) = dz * z * (1 - g)
__device__ __forceinline__ void wn_gate_bwd(float dz, float g, float z, float& duf, float& dug) {
  const float a = g > 1e-30f ? z / g : 0.f;
  duf = dz * g * (1.f - a * a);
  dug = dz * z * (1.f - g);
}

// the two halves separately (same arithmetic as wn_gate_bwd)
__device__ __forceinline__ float wn_gate_bwd_f(float dz, float g, float z) {
  const float a = g > 1e-30f ? z / g : 0.f;
  return dz * g * (1.f - a * a);
}
__device__ __forceinline__ float wn_gate_bwd_g(float dz, float g, float z) { return dz * z * (1.f - g); }

// running max-abs of a tensor (non-negative floats order like their bit patterns); the plain read
// first keeps almost every wave off the atomic (one address: contention would serialise them)
__device__ __forceinline__ void wn_absmax_publish(float* slot, float v) {
  if (v > 0.f && v < 3.0e38f && v > *reinterpret_cast<volatile float*>(slot))
    atomicMax(reinterpret_cast<int*>(slot), __float_as_int(v));
}

// forward range guard: like wn_absmax_publish, but values at or beyond the float range (inf, NaN mapped to 3e38 by
// the caller) are recorded too -- the guard has to see them
__device__ __forceinline__ void wn_absmax_publish_any(float* slot, float v) {
  if (v > 0.f && v > *reinterpret_cast<volatile float*>(slot))
    atomicMax(reinterpret_cast<int*>(slot), __float_as_int(v));
}

// max over |.| taken on the bit patterns: NaN (0x7fc00000) > inf > every finite value, so unlike fmaxf a NaN or an inf among
// the operands is never dropped (wmax >= 0 always, so its own bits order the same way)
__device__ __forceinline__ float wn_absmax_acc(float wmax, float a, float b, float c, float d) {
  const unsigned m0 = max(__float_as_uint(a) & 0x7fffffffu, __float_as_uint(b) & 0x7fffffffu);
  const unsigned m1 = max(__float_as_uint(c) & 0x7fffffffu, __float_as_uint(d) & 0x7fffffffu);
  return __uint_as_float(max(__float_as_uint(wmax), max(m0, m1)));
}
// wave-wide form of the same maximum; non-finite results are clamped to 3e38 ("beyond any limit")
__device__ __forceinline__ float wn_wave_absmax_bits(float wmax) {
  unsigned m = __float_as_uint(wmax);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  const float v = __uint_as_float(m);
  return v < 3.0e38f ? v : 3.0e38f;
}

// Weight image(s) global -> LDS at the top of a kernel whose weights stay resident: EVERY 16-byte load of the calling thread
// is in flight before its first LDS store.  (Written as `for (i = tid; i < n; i += THREADS) dst[i] = src[i]` per image, hipcc
// batches the first seven loads and then runs the rest -- the eighth piece, every piece of a second image, each bias vector --
// as dependent load -> s_waitcnt vmcnt(0) -> ds_write round trips: six global round trips in front of the first tile of
// the fused forward block kernel, ~6 of its 35 us per launch.)  N16_* = compile-time upper bounds of the 16-byte pieces.
template <int THREADS, int MAX1, int MAX2 = 0>
__device__ __forceinline__ void wn_images_to_lds(const void* src1, void* dst1, int n16_1, const void* src2, void* dst2, int n16_2,
                                                 int tid) {
  constexpr int K1 = (MAX1 + THREADS - 1) / THREADS, K2 = (MAX2 + THREADS - 1) / THREADS;
  const f32x4* s1 = reinterpret_cast<const f32x4*>(src1);
  const f32x4* s2 = reinterpret_cast<const f32x4*>(src2);
  f32x4 v1[K1 > 0 ? K1 : 1], v2[K2 > 0 ? K2 : 1];
#pragma unroll
  for (int k = 0; k < K1; ++k) { const int i = tid + k * THREADS; if (i < n16_1) v1[k] = s1[i]; }
#pragma unroll
  for (int k = 0; k < K2; ++k) { const int i = tid + k * THREADS; if (i < n16_2) v2[k] = s2[i]; }
  __builtin_amdgcn_sched_barrier(0);                   // keep the loads in front of the stores
  f32x4* d1 = reinterpret_cast<f32x4*>(dst1);
  f32x4* d2 = reinterpret_cast<f32x4*>(dst2);
#pragma unroll
  for (int k = 0; k < K1; ++k) { const int i = tid + k * THREADS; if (i < n16_1) d1[i] = v1[k]; }
#pragma unroll
  for (int k = 0; k < K2; ++k) { const int i = tid + k * THREADS; if (i < n16_2) d2[i] = v2[k]; }
}

// Keras clips probabilities to [eps, 1 - eps] before the cross entropy (backend.epsilon())
#define WN_KERAS_EPS 1e-7f

// forward activations at or beyond this magnitude trip the range guard of the split-precision mode (fp16 max 65504)
#define WN_RANGE_LIMIT 30000.0f

// Guard publish of the per-sample generation kernels: a lane raises the slot only when ITS running max-abs reached the limit
// (or is inf / NaN) -- no wave reduction and no read of the slot, so it can run every step; below the limit the slot keeps
// whatever the priming pass left there.
__device__ __forceinline__ void wn_guard_publish_over(float* slot, float wmax) {
  const unsigned m = __float_as_uint(wmax);
  if (m >= __float_as_uint(WN_RANGE_LIMIT)) {
    const float v = __uint_as_float(m);
    atomicMax(reinterpret_cast<int*>(slot), __float_as_int(v < 3.0e38f ? v : 3.0e38f));
  }
}

void wn_set_error(const char* fmt, ...);
int wn_debug_get(int key);   // per-thread switches (wn_error.cpp): 1 = exact-fp32 kernels, 9 = no side stream, 24 / 29 = profiling hooks

// XCD-aware walk over the 32-row tiles of a launch.  Workgroups are dealt to the 8 XCDs round-robin by index and every
// XCD has an L2 of its own: with tile = workgroup * waves + wave, a tile's older tap (rows t - d of the same utterance,
// d / 32 tiles back) was fetched through ANOTHER XCD's L2 for every dilation >= 32 and comes from the memory side a second
// time.  Here each XCD takes one contiguous eighth of the tiles, so the rows a tile shares with its neighbours in time
// were requested by a CU behind the same L2.  (Results do not depend on the order of the tiles.)
struct WnTileWalk { int64_t first, end, stride; };
__device__ __forceinline__ WnTileWalk wn_tile_walk(int64_t ntiles, int waves, int wave) {
  WnTileWalk w;
  if ((gridDim.x & 7u) == 0u) {
    const int64_t n8 = (ntiles + 7) >> 3;
    const int64_t x = blockIdx.x & 7u;
    w.first = x * n8 + (int64_t)(blockIdx.x >> 3) * waves + wave;
    w.end = (x + 1) * n8 < ntiles ? (x + 1) * n8 : ntiles;
    w.stride = (int64_t)(gridDim.x >> 3) * waves;
  } else {
    w.first = (int64_t)blockIdx.x * waves + wave;
    w.end = ntiles;
    w.stride = (int64_t)gridDim.x * waves;
  }
  return w;
}
