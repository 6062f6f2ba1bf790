// Elementwise, loss, sampling and reduction kernels of the WaveNet hot path (gfx950).
// References: quantiser src/model.py:151-153; mu-law src/utils.py:34-35; inverse
// src/callbacks.py:126-131; losses src/model.py:505-551; samplers src/model.py:393-503.
#include <algorithm>

#include "wn_kernels.h"
#include "wn_sample.h"


static inline int wn_blocks(int64_t n, int per = 256, int cap = 4096) {
  int64_t b = (n + per - 1) / per;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------
__global__ void wn_add_kernel(const float* a, const float* b, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}
int wn_launch_add(const float* a, const float* b, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_add_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, a, b, out, n);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_fill_kernel(float* p, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
int wn_launch_fill(float* p, float v, int64_t n, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_fill_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, p, v, n);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// out[b][c] = sum_t g[b][t][c]: stage 1 sums WN_CS_CHUNKS time chunks per utterance in parallel
// (one block per (utterance, chunk, 64 channels)), stage 2 folds the chunks in a fixed order.
#define WN_CS_CHUNKS 128
__global__ void wn_colsum_stage1(const float* g, int T, int C, float* part) {
  __shared__ float sm[256];
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int c = blockIdx.z * 64 + (threadIdx.x & 63);
  const int p = threadIdx.x >> 6;
  const int len = (T + WN_CS_CHUNKS - 1) / WN_CS_CHUNKS;
  const int t0 = chunk * len, t1 = min(T, t0 + len);
  float acc = 0.f;
  if (c < C)
    for (int t = t0 + p; t < t1; t += 4) acc += g[((int64_t)b * T + t) * C + c];
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (p == 0 && c < C)
    part[((int64_t)b * WN_CS_CHUNKS + chunk) * C + c] = sm[threadIdx.x] + sm[threadIdx.x + 64] + sm[threadIdx.x + 128] + sm[threadIdx.x + 192];
}
__global__ void wn_colsum_stage2(const float* part, int C, float* out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / C;
  const int c = (int)(i % C);
  double acc = 0.0;
  for (int k = 0; k < WN_CS_CHUNKS; ++k) acc += (double)part[(b * WN_CS_CHUNKS + k) * C + c];
  out[i] = (float)acc;
}
// scratch: B * WN_CS_CHUNKS * C floats
int64_t wn_colsum_scratch_floats(int B, int C) { return (int64_t)B * WN_CS_CHUNKS * C; }
int wn_launch_colsum_per_batch(const float* g, int B, int T, int C, float* out, float* scratch, hipStream_t s) {
  hipLaunchKernelGGL(wn_colsum_stage1, dim3(B, WN_CS_CHUNKS, (C + 63) / 64), dim3(256), 0, s, g, T, C, scratch);
  const int64_t total = (int64_t)B * C;
  hipLaunchKernelGGL(wn_colsum_stage2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, scratch, C, out, total);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// quantiser: index = #{ j in 1..2^bits-1 : edge_j <= x },  edge_j = -1 + j * 2^(1-bits) (exact in
// fp32).  Candidate from arithmetic, then corrected against the exact edges so that the result is
// the upper-bound search of tf Bucketize bit for bit.
__device__ __forceinline__ int wn_quantize_one(float x, int bits) {
  const int nmax = (1 << bits) - 1;
  const double step = ldexp(1.0, 1 - bits);
  double f = floor(((double)x + 1.0) / step);
  int idx = f < 0.0 ? 0 : (f > (double)nmax ? nmax : (int)f);
  while (idx > 0 && (float)(-1.0 + idx * step) > x) --idx;
  while (idx < nmax && (float)(-1.0 + (idx + 1) * step) <= x) ++idx;
  return idx;
}
__global__ void wn_quantize_kernel(const float* x, int32_t* idx, int64_t n, int bits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    idx[i] = wn_quantize_one(x[i], bits);
}
int wn_launch_quantize(const float* x, int32_t* idx, int64_t n, int bits, hipStream_t s) {
  if (n <= 0) return WN_OK;
  if (bits < 1 || bits > 16) { wn_set_error("quantize: bits %d unsupported", bits); return WN_E_UNSUPPORTED; }
  hipLaunchKernelGGL(wn_quantize_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, x, idx, n, bits);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_dequantize_kernel(const int32_t* idx, float* x, int64_t n, float inv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] = (float)idx[i] * inv - 1.0f;       // i / 2^(bits-1) - 1  (src/model.py:411,418)
}
int wn_launch_dequantize(const int32_t* idx, float* x, int64_t n, int bits, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_dequantize_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, idx, x, n,
                     1.0f / (float)(1 << (bits - 1)));
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_mulaw_kernel(const float* x, float* y, int64_t n) {
  const float inv = 1.0f / logf(256.0f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float sgn = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
    y[i] = sgn * (logf(1.0f + 255.0f * fabsf(v)) * inv);
  }
}
int wn_launch_mulaw(const float* x, float* y, int64_t n, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_mulaw_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, x, y, n);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_inv_mulaw_kernel(const float* y, float* x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = y[i];
    const float sgn = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
    x[i] = sgn * (powf(256.0f, fabsf(v)) - 1.0f) / 255.0f;
  }
}
int wn_launch_inv_mulaw(const float* y, float* x, int64_t n, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_inv_mulaw_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, y, x, n);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ __launch_bounds__(256) void wn_softmax_kernel(const float* logits, float* probs,
                                                         int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* l = logits + row * C;
  float m = -INFINITY;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, l[j]);
  m = wn_wave_max(m);
  float z = 0.f;
  for (int j = lane; j < C; j += 64) z += expf(l[j] - m);
  z = wn_wave_sum(z);
  const float inv = 1.0f / z;
  for (int j = lane; j < C; j += 64) probs[row * C + j] = expf(l[j] - m) * inv;
}
int wn_launch_softmax(const float* logits, float* probs, int64_t rows, int C, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits,
                     probs, rows, C);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// Keras sparse_categorical_crossentropy(target, softmax(logits)), from_logits=False:
//   q = softmax(logits); p = clip(q, eps, 1-eps); loss = -(log p_t - log sum_j p_j)
// and its gradient w.r.t. the logits (clip passes gradient where eps <= q <= 1-eps).
// sample_out (C <= 256 only): also draw sample_waveform(softmax(logits)) of the row (src/model.py:338,407-411)
// from the probabilities already in registers -- the values wn_softmax_kernel would write, the draw
// wn_sample_rand_cat_kernel would make from them.
// C <= 256, the shape of every BASELINE categorical head: persistent waves, one row at a time per wave with the NEXT row's
// logits and target already requested (the one-row-per-wave launch below spends most of a row waiting for its loads:
// 172 us for 268 MB).  Per row the arithmetic is exactly that of wn_cat_loss_kernel's C <= 256 branch -- the target's
// probability comes from the lane that holds it instead of a second, dependent load of the same logit -- and the max-abs
// of the gradients is published once per wave.
__global__ __launch_bounds__(256) void wn_cat_loss256_kernel(const float* logits, const int32_t* target,
                                                             int64_t rows, int C, float gscale,
                                                             float* loss_rows, float* g_logits, float* absmax_out,
                                                             float* sample_out, float inv_lv, uint64_t seed, uint64_t offset) {
  __shared__ float qs[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + w;
  float vn[4];
  int tn = 0;
  auto fetch = [&](int64_t r) {
    const float* l = logits + r * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) vn[k] = lane + 64 * k < C ? l[lane + 64 * k] : -INFINITY;
    tn = target[r];
  };
  if (row < rows) fetch(row);
  float gmax = 0.f;
  for (; row < rows; row += stride) {
    float v[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = vn[k];
    int tgt = tn;
    if (row + stride < rows) fetch(row + stride);     // in flight while this row is worked on
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(m, v[k]);
    m = wn_wave_max(m);
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      e[k] = lane + 64 * k < C ? expf(v[k] - m) : 0.f;
      if (lane + 64 * k < C) z += e[k];
    }
    z = wn_wave_sum(z);
    const float inv = 1.0f / z;
    if (sample_out) {
      float* qw = qs[w];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (lane + 64 * k < C) qw[lane + 64 * k] = e[k] * inv;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int drawn = wn_draw_cat_row((const float*)qw, C, lane, row, seed, offset);
      if (lane == 0) sample_out[row] = (float)drawn * inv_lv - 1.0f;
      __builtin_amdgcn_wave_barrier();                // the next row rewrites qw
    }
    float S = 0.f, A = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < C) {
        const float q = e[k] * inv;
        S += fminf(fmaxf(q, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
        if (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) A += q;
      }
    S = wn_wave_sum(S);
    A = wn_wave_sum(A);
    tgt = tgt < 0 ? 0 : (tgt >= C ? C - 1 : tgt);
    const int tk = tgt >> 6;                          // wave-uniform: the lane tgt & 63 holds e[tk] = exp(l[tgt] - m)
    const float et = __shfl(tk == 0 ? e[0] : tk == 1 ? e[1] : tk == 2 ? e[2] : e[3], tgt & 63);
    const float qt = et * inv;
    const float pt = fminf(fmaxf(qt, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
    const float ct = (qt >= WN_KERAS_EPS && qt <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
    if (lane == 0) loss_rows[row] = -(logf(pt) - logf(S));
    if (g_logits) {
      const float invS = 1.0f / S;
      const float dot = A * invS - ct * qt / pt;     // sum_j g_j q_j
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        if (j < C) {
          const float q = e[k] * inv;
          const float c = (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
          float g = c * invS;
          if (j == tgt) g -= ct / pt;
          const float gl = gscale * q * (g - dot);
          g_logits[row * C + j] = gl;
          gmax = fmaxf(gmax, fabsf(gl));
        }
      }
    }
  }
  if (g_logits && absmax_out) {
    gmax = wn_wave_max(gmax);
    if (lane == 0) wn_absmax_publish(absmax_out, gmax);
  }
}

__global__ __launch_bounds__(256) void wn_cat_loss_kernel(const float* logits, const int32_t* target,
                                                          int64_t rows, int C, float gscale,
                                                          float* loss_rows, float* g_logits, float* absmax_out,
                                                          float* sample_out, float inv_lv, uint64_t seed, uint64_t offset) {
  __shared__ float qs[4][256];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* l = logits + row * C;
  if (C <= 256) {
    // the row lives in registers (element k of a lane = class lane + 64 k, the same assignment and the
    // same per-lane summation order as the general path below): one read of the logits, one exp per class
    float v[4], e[4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = lane + 64 * k;
      v[k] = j < C ? l[j] : -INFINITY;
      m = fmaxf(m, v[k]);
    }
    m = wn_wave_max(m);
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      e[k] = lane + 64 * k < C ? expf(v[k] - m) : 0.f;
      if (lane + 64 * k < C) z += e[k];
    }
    z = wn_wave_sum(z);
    const float inv = 1.0f / z;
    if (sample_out) {
      float* qw = qs[threadIdx.x >> 6];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (lane + 64 * k < C) qw[lane + 64 * k] = e[k] * inv;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int drawn = wn_draw_cat_row((const float*)qw, C, lane, row, seed, offset);
      if (lane == 0) sample_out[row] = (float)drawn * inv_lv - 1.0f;
    }
    float S = 0.f, A = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < C) {
        const float q = e[k] * inv;
        S += fminf(fmaxf(q, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
        if (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) A += q;
      }
    S = wn_wave_sum(S);
    A = wn_wave_sum(A);
    int tgt = target[row];
    tgt = tgt < 0 ? 0 : (tgt >= C ? C - 1 : tgt);
    const float qt = expf(l[tgt] - m) * inv;
    const float pt = fminf(fmaxf(qt, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
    const float ct = (qt >= WN_KERAS_EPS && qt <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
    if (lane == 0) loss_rows[row] = -(logf(pt) - logf(S));
    if (g_logits) {
      float gmax = 0.f;
      const float invS = 1.0f / S;
      const float dot = A * invS - ct * qt / pt;     // sum_j g_j q_j
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = lane + 64 * k;
        if (j < C) {
          const float q = e[k] * inv;
          const float c = (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
          float g = c * invS;
          if (j == tgt) g -= ct / pt;
          const float gl = gscale * q * (g - dot);
          g_logits[row * C + j] = gl;
          gmax = fmaxf(gmax, fabsf(gl));
        }
      }
      if (absmax_out) {
        gmax = wn_wave_max(gmax);
        if (lane == 0) wn_absmax_publish(absmax_out, gmax);
      }
    }
    return;
  }
  float m = -INFINITY;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, l[j]);
  m = wn_wave_max(m);
  float z = 0.f;
  for (int j = lane; j < C; j += 64) z += expf(l[j] - m);
  z = wn_wave_sum(z);
  const float inv = 1.0f / z;
  float S = 0.f, A = 0.f;
  for (int j = lane; j < C; j += 64) {
    const float q = expf(l[j] - m) * inv;
    const float p = fminf(fmaxf(q, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
    S += p;
    if (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) A += q;
  }
  S = wn_wave_sum(S);
  A = wn_wave_sum(A);
  int tgt = target[row];
  tgt = tgt < 0 ? 0 : (tgt >= C ? C - 1 : tgt);
  const float qt = expf(l[tgt] - m) * inv;
  const float pt = fminf(fmaxf(qt, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
  const float ct = (qt >= WN_KERAS_EPS && qt <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
  if (lane == 0) loss_rows[row] = -(logf(pt) - logf(S));
  if (g_logits) {
    float gmax = 0.f;
    const float invS = 1.0f / S;
    const float dot = A * invS - ct * qt / pt;     // sum_j g_j q_j
    for (int j = lane; j < C; j += 64) {
      const float q = expf(l[j] - m) * inv;
      const float c = (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
      float g = c * invS;
      if (j == tgt) g -= ct / pt;
      const float gl = gscale * q * (g - dot);
      g_logits[row * C + j] = gl;
      gmax = fmaxf(gmax, fabsf(gl));
    }
    if (absmax_out) {
      gmax = wn_wave_max(gmax);
      if (lane == 0) wn_absmax_publish(absmax_out, gmax);
    }
  }
}
int wn_launch_cat_loss(const float* logits, const int32_t* target, int64_t rows, int C,
                       float gscale, float* loss_rows, float* g_logits, float* absmax_out, hipStream_t s,
                       float* sample_out, int bits, uint64_t seed, uint64_t offset) {
  if (rows <= 0) return WN_OK;
  if (sample_out && C > 256) { wn_set_error("cat_loss: the in-kernel sample draw needs <= 256 classes"); return WN_E_UNSUPPORTED; }
  const float inv_lv = sample_out ? 1.0f / (float)(1 << (bits - 1)) : 0.f;
  if (C <= 256) {
    const int64_t wgs = std::min<int64_t>((rows + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(wn_cat_loss256_kernel, dim3((unsigned)wgs), dim3(256), 0, s, logits, target, rows, C, gscale,
                       loss_rows, g_logits, absmax_out, sample_out, inv_lv, seed, offset);
  } else {
    hipLaunchKernelGGL(wn_cat_loss_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits,
                       target, rows, C, gscale, loss_rows, g_logits, absmax_out, sample_out, inv_lv, seed, offset);
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ __launch_bounds__(256) void wn_cat_loss_probs_kernel(const float* probs,
                                                                const int32_t* target, int64_t rows,
                                                                int C, float* loss_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* q = probs + row * C;
  float S = 0.f;
  for (int j = lane; j < C; j += 64) S += fminf(fmaxf(q[j], WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
  S = wn_wave_sum(S);
  int tgt = target[row];
  tgt = tgt < 0 ? 0 : (tgt >= C ? C - 1 : tgt);
  const float pt = fminf(fmaxf(q[tgt], WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
  if (lane == 0) loss_rows[row] = -(logf(pt) - logf(S));
}
int wn_launch_cat_loss_probs(const float* probs, const int32_t* target, int64_t rows, int C,
                             float* loss_rows, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_cat_loss_probs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                     probs, target, rows, C, loss_rows);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// mixture losses, one thread per (b,t) row; M <= 32.  Evaluated in double: with bits = 16 the
// half-bin (src/model.py:538) is 7.6e-6, so sigmoid(a) - sigmoid(b) cancels ~5 digits and an
// fp32 evaluation (the reference's own included) carries 1e-3..1e-2 relative noise per term.
#define WN_MAXMIX 32
__device__ __forceinline__ double wn_sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }
__global__ void wn_mix_loss_kernel(const float* pred, const float* y, int64_t rows, int M, int bits,
                                   int kind, float gscale, float* loss_rows, float* g_pred, float* absmax_out) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float* p = pred + row * 3 * M;
  const double yy = (double)y[row];
  double w[WN_MAXMIX], comp[WN_MAXMIX];
  double wm = -INFINITY;
  for (int k = 0; k < M; ++k) wm = fmax(wm, (double)p[k]);
  double wz = 0.0;
  for (int k = 0; k < M; ++k) { w[k] = exp((double)p[k] - wm); wz += w[k]; }
  const double winv = 1.0 / wz;
  const double halfbit = 0.5 / (double)(1 << bits);                      // src/model.py:538
  const double sqrt2pi = sqrt(2.0 * 3.14159265359);                      // src/model.py:9
  double lik = 0.0;
  for (int k = 0; k < M; ++k) {
    w[k] *= winv;
    const double mu = (double)p[M + k];
    const double ls = fmax((double)p[2 * M + k], -7.0);
    if (kind == 1) {
      const double inv = exp(-ls);
      comp[k] = wn_sigmoid_d((yy - mu + halfbit) * inv) - wn_sigmoid_d((yy - mu - halfbit) * inv);
    } else {
      const double sc = exp(ls);
      const double xx = fmin((yy - mu) / sc, 1e8);
      comp[k] = exp(-0.5 * xx * xx) / (sc * sqrt2pi);
    }
    lik += w[k] * comp[k];
  }
  loss_rows[row] = (float)(-log(lik));
  if (!g_pred) return;
  float* g = g_pred + row * 3 * M;
  const double dl = -(double)gscale / lik;                               // dL/dlik
  for (int k = 0; k < M; ++k) {
    const double mu = (double)p[M + k];
    const double lsr = (double)p[2 * M + k];
    const double ls = fmax(lsr, -7.0);
    const double lsmask = lsr >= -7.0 ? 1.0 : 0.0;
    g[k] = (float)(dl * (w[k] * comp[k] - w[k] * lik));
    if (kind == 1) {
      const double inv = exp(-ls);
      const double a = (yy - mu + halfbit) * inv, b = (yy - mu - halfbit) * inv;
      const double sa = wn_sigmoid_d(a), sb = wn_sigmoid_d(b);
      const double da = sa * (1.0 - sa), db = sb * (1.0 - sb);
      g[M + k] = (float)(dl * (-w[k] * inv * (da - db)));
      g[2 * M + k] = (float)(dl * lsmask * (-w[k] * (a * da - b * db)));
    } else {
      const double sc = exp(ls);
      const double xr = (yy - mu) / sc;
      const double xx = fmin(xr, 1e8);
      const double xmask = xr <= 1e8 ? 1.0 : 0.0;
      const double pdf = comp[k];
      // d pdf/d mu = pdf * xx / sc ; d pdf/d ls = pdf * (xx^2 - 1)
      g[M + k] = (float)(dl * w[k] * pdf * xx / sc * xmask);
      g[2 * M + k] = (float)(dl * lsmask * w[k] * pdf * (xx * xx * xmask - 1.0));
    }
  }
  if (absmax_out) {
    float gmax = 0.f;
    for (int k = 0; k < 3 * M; ++k) gmax = fmaxf(gmax, fabsf(g[k]));
    wn_absmax_publish(absmax_out, gmax);
  }
}
int wn_launch_mix_loss(const float* pred, const float* y, int64_t rows, int M, int bits, int kind,
                       float gscale, float* loss_rows, float* g_pred, float* absmax_out, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  if (M < 1 || M > WN_MAXMIX) { wn_set_error("mix_loss: num_mixtures %d unsupported (max %d)", M, WN_MAXMIX); return WN_E_UNSUPPORTED; }
  hipLaunchKernelGGL(wn_mix_loss_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, pred,
                     y, rows, M, bits, kind, gscale, loss_rows, g_pred, absmax_out);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// deterministic two-stage sum (double accumulation), out[0] = scale * sum(v)
__global__ void wn_sum_stage1(const float* v, int64_t n, double* scratch) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    acc += (double)v[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scratch[blockIdx.x] = sm[0];
}
__global__ void wn_sum_stage2(const double* scratch, int nb, float scale, float* out) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += scratch[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sm[0] * (double)scale);
}
// scratch: >= 1024 doubles (8 KiB)
int wn_launch_sum(const float* v, int64_t n, float scale, float* out, float* scratch, hipStream_t s) {
  int nb = wn_blocks(n, 256, 1024);
  hipLaunchKernelGGL(wn_sum_stage1, dim3(nb), dim3(256), 0, s, v, n, reinterpret_cast<double*>(scratch));
  hipLaunchKernelGGL(wn_sum_stage2, dim3(1), dim3(256), 0, s, reinterpret_cast<const double*>(scratch), nb, scale, out);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// out[0] = scale * sum((a - b)^2): tf.keras.metrics.MeanSquaredError(y_true, sample) of a step (src/model.py:346,
// train.py:227) with scale = 1 / (n * replicas); same two-stage double accumulation as wn_launch_sum
__global__ void wn_sqdiff_stage1(const float* a, const float* b, int64_t n, double* scratch) {
  __shared__ double sm[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    acc += (double)(d * d);
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scratch[blockIdx.x] = sm[0];
}
int wn_launch_sqdiff_sum(const float* a, const float* b, int64_t n, float scale, float* out, float* scratch, hipStream_t s) {
  int nb = wn_blocks(n, 256, 1024);
  hipLaunchKernelGGL(wn_sqdiff_stage1, dim3(nb), dim3(256), 0, s, a, b, n, reinterpret_cast<double*>(scratch));
  hipLaunchKernelGGL(wn_sum_stage2, dim3(1), dim3(256), 0, s, reinterpret_cast<const double*>(scratch), nb, scale, out);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// samplers
// (queued generation: the sample also goes to its place in the output rows and into the network's input ring -- the
// emit step of a generation step rides in the sampler's launch)
__device__ __forceinline__ void wn_emit_sample(const WnEmit& e, int64_t row, float v) {
  if (!e.out) return;
  e.out[row * e.length + e.step] = v;
  if (e.xin_slot) e.xin_slot[row] = v;
}
__global__ __launch_bounds__(256) void wn_sample_det_cat_kernel(const float* pred, int64_t rows, int C,
                                                                float inv, float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = pred + row * C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < C; j += 64) {
    const float v = p[j];
    if (v > best) { best = v; bi = j; }          // strictly greater keeps the first maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = (float)bi * inv - 1.0f;
}
__global__ void wn_sample_det_mix_kernel(const float* pred, int64_t rows, int M, float* out, WnEmit em) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float v = wn_mix_det_row(pred + row * 3 * M, M);
  out[row] = v;
  wn_emit_sample(em, row, v);
}
int wn_launch_sample_det(const float* pred, int64_t rows, int C, int M, int bits, float* out,
                         hipStream_t s) {
  return wn_launch_sample_det_emit(pred, rows, C, M, bits, out, WnEmit{nullptr, 0, 0, nullptr}, s);
}
// (categorical rows with an emit target go through wn_launch_gen_tail_cat_det, which starts from the logits)
int wn_launch_sample_det_emit(const float* pred, int64_t rows, int C, int M, int bits, float* out, WnEmit em, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  if (M <= 0 && em.out) { wn_set_error("sample_det_emit: categorical rows use the fused tail"); return WN_E_INVALID; }
  if (M <= 0) {
    hipLaunchKernelGGL(wn_sample_det_cat_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                       pred, rows, C, 1.0f / (float)(1 << (bits - 1)), out);
  } else {
    hipLaunchKernelGGL(wn_sample_det_mix_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       s, pred, rows, M, out, em);
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ __launch_bounds__(256) void wn_sample_rand_cat_kernel(const float* pred, int64_t rows, int C,
                                                                 float inv, uint64_t seed, uint64_t offset,
                                                                 float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int result = wn_draw_cat_row(pred + row * C, C, lane, row, seed, offset);
  if (lane == 0) out[row] = (float)result * inv - 1.0f;
}

// The same draw straight from the logits (training step with a compiled sample metric, src/model.py:338):
// the probabilities are those of wn_softmax_kernel (same lane assignment, same reductions), kept in LDS
// instead of a (rows, C) tensor in HBM, so the drawn class is the one sample_waveform(softmax(logits)) draws.
#define WN_SAMPLE_FUSED_MAXC 1024
__global__ __launch_bounds__(256) void wn_sample_rand_cat_logits_kernel(const float* logits, int64_t rows, int C,
                                                                        float inv_lv, uint64_t seed, uint64_t offset,
                                                                        float* out, WnEmit em) {
  __shared__ float q[4][WN_SAMPLE_FUSED_MAXC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  if (row >= rows) return;
  const float v = wn_cat_rand_row(logits + row * C, C, lane, q[w], row, seed, offset, inv_lv);
  if (lane == 0) {
    out[row] = v;
    wn_emit_sample(em, row, v);
  }
}
int wn_sample_from_logits_supported(int C) { return C <= WN_SAMPLE_FUSED_MAXC ? 1 : 0; }
int wn_launch_sample_rand_cat_logits(const float* logits, int64_t rows, int C, int bits, uint64_t seed, uint64_t offset,
                                     float* out, hipStream_t s) {
  return wn_launch_sample_rand_cat_logits_emit(logits, rows, C, bits, seed, offset, out, WnEmit{nullptr, 0, 0, nullptr}, s);
}
int wn_launch_sample_rand_cat_logits_emit(const float* logits, int64_t rows, int C, int bits, uint64_t seed, uint64_t offset,
                                          float* out, WnEmit em, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  if (C > WN_SAMPLE_FUSED_MAXC) { wn_set_error("sample from logits: %d classes > %d", C, WN_SAMPLE_FUSED_MAXC); return WN_E_UNSUPPORTED; }
  hipLaunchKernelGGL(wn_sample_rand_cat_logits_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, rows, C,
                     1.0f / (float)(1 << (bits - 1)), seed, offset, out, em);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
__global__ void wn_sample_rand_mix_kernel(const float* pred, int64_t rows, int M, int kind, uint64_t seed,
                                          uint64_t offset, float* out, WnEmit em) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float vc = wn_mix_rand_row(pred + row * 3 * M, M, kind, row, seed, offset);
  out[row] = vc;
  wn_emit_sample(em, row, vc);
}
int wn_launch_sample_rand(const float* pred, int64_t rows, int C, int M, int bits, int kind,
                          uint64_t seed, uint64_t offset, float* out, hipStream_t s) {
  return wn_launch_sample_rand_emit(pred, rows, C, M, bits, kind, seed, offset, out, WnEmit{nullptr, 0, 0, nullptr}, s);
}
// (categorical rows with an emit target go through wn_launch_sample_rand_cat_logits_emit)
int wn_launch_sample_rand_emit(const float* pred, int64_t rows, int C, int M, int bits, int kind,
                               uint64_t seed, uint64_t offset, float* out, WnEmit em, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  if (M <= 0 && em.out) { wn_set_error("sample_rand_emit: categorical rows start from the logits"); return WN_E_INVALID; }
  if (M <= 0) {
    hipLaunchKernelGGL(wn_sample_rand_cat_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                       pred, rows, C, 1.0f / (float)(1 << (bits - 1)), seed, offset, out);
  } else {
    hipLaunchKernelGGL(wn_sample_rand_mix_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       s, pred, rows, M, kind, seed, offset, out, em);
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// optimizer: per-tensor clipnorm + Keras Adam  (train.py:225-226, src/model.py:336)
__global__ void wn_sumsq_kernel(const float* g, const WnTensorDesc* table, float* norms2) {
  __shared__ double sm[256];
  const WnTensorDesc d = table[blockIdx.x];
  const float* p = g + d.off;
  // four independent chains: a single one ran at one memory round trip per element (63 us for 1.25 M values)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t i = threadIdx.x;
  for (; i + 3 * 256 < d.len; i += 4 * 256) {
    const float v0 = p[i], v1 = p[i + 256], v2 = p[i + 512], v3 = p[i + 768];
    a0 += (double)v0 * (double)v0; a1 += (double)v1 * (double)v1;
    a2 += (double)v2 * (double)v2; a3 += (double)v3 * (double)v3;
  }
  for (; i < d.len; i += 256) a0 += (double)p[i] * (double)p[i];
  const double acc = (a0 + a1) + (a2 + a3);
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) norms2[blockIdx.x] = (float)sm[0];
}
int wn_launch_sumsq(const float* g, const WnTensorDesc* d_table, int n, float* norms2, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_sumsq_kernel, dim3(n), dim3(256), 0, s, g, d_table, norms2);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_adam_kernel(float* p, const float* g, float* m, float* v, const WnTensorDesc* table,
                               const float* norms2, float clipnorm, float alpha, float beta1,
                               float beta2, float eps, const float* skip_flag) {
  if (skip_flag && *skip_flag != 0.f) return;     // range guard tripped: the step is redone in exact fp32 (uniform)
  const WnTensorDesc d = table[blockIdx.y];
  float scale = 1.0f;
  if (clipnorm > 0.f) {
    const float nrm = sqrtf(norms2[blockIdx.y]);
    scale = clipnorm / fmaxf(nrm, clipnorm);     // tf.clip_by_norm
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = d.off + i;
    const float gi = g[k] * scale;
    const float mi = m[k] + (gi - m[k]) * (1.0f - beta1);
    const float vi = v[k] + (gi * gi - v[k]) * (1.0f - beta2);
    m[k] = mi;
    v[k] = vi;
    p[k] = p[k] - alpha * mi / (sqrtf(vi) + eps);
  }
}
int wn_launch_adam(float* p, const float* g, float* m, float* v, const WnTensorDesc* d_table, int n,
                   const float* norms2, float clipnorm, float alpha, float beta1, float beta2,
                   float eps, const float* skip_flag, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_adam_kernel, dim3(32, n), dim3(256), 0, s, p, g, m, v, d_table, norms2,
                     clipnorm, alpha, beta1, beta2, eps, skip_flag);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// gate of the composed (unfused) block path: u (rows, 2D) -> a = tanh(u[:, :D]), g = sigmoid(u[:, D:]),
// z = a * g   (src/layers.py:208-210)
__global__ void wn_gate_kernel(const float* u, int64_t rows, int D, float* ag, float* z, int ldz) {
  const int64_t total = rows * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int c = (int)(i % D);
    const float a = wn_tanh_fast(u[r * 2 * D + c]);
    const float g = wn_sigmoid_fast(u[r * 2 * D + D + c]);
    if (ag) ag[r * D + c] = g;          // saved sigmoid only (tanh = z / sigmoid in backward)
    if (z) z[r * ldz + c] = a * g;
  }
}
int wn_launch_gate(const float* u, int64_t rows, int D, float* ag, float* z, int ldz, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_gate_kernel, dim3(wn_blocks(rows * D)), dim3(256), 0, s, u, rows, D, ag, z, ldz);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// out[b][n] = sum_sp slab[b][sp][n]
__global__ void wn_batch_reduce_kernel(const float* slab, int splits, int N, float* out, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / N;
  const int n = (int)(i % N);
  float acc = 0.f;
  for (int s = 0; s < splits; ++s) acc += slab[(b * splits + s) * N + n];
  out[i] = acc;
}
int wn_launch_batch_reduce(const float* slab, int B, int splits, int N, float* out, hipStream_t s) {
  const int64_t total = (int64_t)B * N;
  if (total <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_batch_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     slab, splits, N, out, total);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// y[off + i] += coef * x[off + i] over the tensors of a table (L2 regulariser gradient)
__global__ void wn_axpy_table_kernel(float* y, const float* x, const WnTensorDesc* table, float coef) {
  const WnTensorDesc d = table[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.len;
       i += (int64_t)gridDim.x * blockDim.x)
    y[d.off + i] += coef * x[d.off + i];
}
int wn_launch_axpy_table(float* y, const float* x, const WnTensorDesc* d_table, int n, float coef,
                         hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_axpy_table_kernel, dim3(16, n), dim3(256), 0, s, y, x, d_table, coef);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// out = g * act'(y)   (y = saved activation output)
__global__ void wn_dact_mul_kernel(const float* g, const float* y, float* out, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = g[i] * wn_dact_from_y(y[i], act);
}
int wn_launch_dact_mul(const float* g, const float* y, float* out, int64_t n, int act, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_dact_mul_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, g, y, out, n, act);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// Dropout on the block input (src/layers.py:108-111,195-196): keep-mask from a counter-based integer
// hash of (seed, block, step, element index) -- reproducible, stateless, identical in forward and
// backward (TF's own random stream is not reproducible; the CPU oracle restates this hash).
//   forward : xd = keep ? x / (1 - rate) : 0
//   backward: g_x = (keep ? g_xd / (1 - rate) : 0) + g_res      (g_res = residual path, may be null)
__device__ __forceinline__ uint32_t wn_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool wn_drop_keep(int64_t idx, uint32_t key, float rate) {
  const uint32_t lo = (uint32_t)idx, hi = (uint32_t)((uint64_t)idx >> 32);
  const uint32_t hsh = wn_hash32(lo ^ wn_hash32(hi + key));
  return (float)(hsh >> 8) * (1.0f / 16777216.0f) >= rate;
}
__global__ void wn_dropout_kernel(const float* x, const float* g_res, float* out, int64_t n, float rate, float scale,
                                  uint32_t key, float* absmax_out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = wn_drop_keep(i, key, rate) ? x[i] * scale : 0.f;
    if (g_res) v += g_res[i];
    out[i] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (absmax_out) {
    m = wn_wave_max(m);
    if ((threadIdx.x & 63) == 0) wn_absmax_publish(absmax_out, m);
  }
}
uint32_t wn_dropout_key(uint64_t seed, int block, uint64_t step) {
  uint32_t k = (uint32_t)seed * 0x9E3779B9U + (uint32_t)(seed >> 32);
  k ^= (uint32_t)block * 0x85EBCA6BU + 0x1234567U;
  k ^= (uint32_t)step * 0xC2B2AE35U + (uint32_t)(step >> 32) * 0x27D4EB2FU;
  return k;
}
int wn_launch_dropout(const float* x, const float* g_res, float* out, int64_t n, float rate, uint32_t key,
                      float* absmax_out, hipStream_t s) {
  if (n <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_dropout_kernel, dim3(wn_blocks(n)), dim3(256), 0, s, x, g_res, out, n, rate, 1.0f / (1.0f - rate),
                     key, absmax_out);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ---------------------------------------------------------------------------------------------
// Global conditioning of ALL residual blocks at once (src/layers.py:116-120,203-204 with the
// time-invariant condition of src/model.py:221-225: conv_cond(repeat(m)) is a per-utterance bias).
// The N per-block 1x1 convs are one contraction tmp[B][N*2D] = m . [W_c^1 | ... | W_c^N]; these three
// kernels move between that layout and the per-block tensors.
// ---------------------------------------------------------------------------------------------
// cb[b][u][n] = tmp[u][b*2D + n] + b_c^b[n]
__global__ void wn_cond_scatter_kernel(const float* tmp, const float* params, int64_t b_off0, int64_t b_stride, int B,
                                       int N, int D2, float* cb) {
  const int64_t total = (int64_t)N * B * D2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % D2);
    const int u = (int)((i / D2) % B);
    const int b = (int)(i / ((int64_t)D2 * B));
    cb[i] = tmp[((int64_t)u * N + b) * D2 + n] + params[b_off0 + (int64_t)b * b_stride + n];
  }
}
// dcb[u][b*2D + n] = sum over the time splits of utterance u of the db_d partial sums that the weight
// gradient kernels left in their slab rows (they ARE the per-utterance column sums of d u)
__global__ void wn_cond_gather_kernel(const float* slab, int64_t P, int spb, int64_t bd_off0, int64_t bd_stride, int B,
                                      int N, int D2, float* dcb) {
  const int64_t total = (int64_t)B * N * D2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % D2);
    const int b = (int)((i / D2) % N);
    const int u = (int)(i / ((int64_t)D2 * N));
    float acc = 0.f;
    for (int sp = 0; sp < spb; ++sp) acc += slab[(int64_t)(u * spb + sp) * P + bd_off0 + (int64_t)b * bd_stride + n];
    dcb[i] = acc;
  }
}
// dW_c^b[c][n] = sum_u m[u][c] dcb[u][b*2D + n];  db_c^b[n] = sum_u dcb[u][b*2D + n]
__global__ void wn_cond_wgrad_kernel(const float* m, const float* dcb, int B, int Cc, int N, int D2, float* grads,
                                     int64_t w_off0, int64_t w_stride, int64_t b_off0, int64_t b_stride) {
  const int64_t total = (int64_t)N * (Cc + 1) * D2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % D2);
    const int c = (int)((i / D2) % (Cc + 1));              // row Cc = the bias
    const int b = (int)(i / ((int64_t)D2 * (Cc + 1)));
    float acc = 0.f;
    for (int u = 0; u < B; ++u) {
      const float g = dcb[((int64_t)u * N + b) * D2 + n];
      acc += c < Cc ? m[(int64_t)u * Cc + c] * g : g;
    }
    if (c < Cc) grads[w_off0 + (int64_t)b * w_stride + (int64_t)c * D2 + n] = acc;
    else grads[b_off0 + (int64_t)b * b_stride + n] = acc;
  }
}
int wn_launch_cond_scatter(const float* tmp, const float* params, int64_t b_off0, int64_t b_stride, int B, int N, int D2,
                           float* cb, hipStream_t s) {
  const int64_t total = (int64_t)N * B * D2;
  hipLaunchKernelGGL(wn_cond_scatter_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1024)), dim3(256), 0, s, tmp,
                     params, b_off0, b_stride, B, N, D2, cb);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
int wn_launch_cond_gather(const float* slab, int64_t P, int spb, int64_t bd_off0, int64_t bd_stride, int B, int N, int D2,
                          float* dcb, hipStream_t s) {
  const int64_t total = (int64_t)B * N * D2;
  hipLaunchKernelGGL(wn_cond_gather_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1024)), dim3(256), 0, s, slab, P,
                     spb, bd_off0, bd_stride, B, N, D2, dcb);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
int wn_launch_cond_wgrad(const float* m, const float* dcb, int B, int Cc, int N, int D2, float* grads, int64_t w_off0,
                         int64_t w_stride, int64_t b_off0, int64_t b_stride, hipStream_t s) {
  const int64_t total = (int64_t)N * (Cc + 1) * D2;
  hipLaunchKernelGGL(wn_cond_wgrad_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1024)), dim3(256), 0, s, m, dcb, B,
                     Cc, N, D2, grads, w_off0, w_stride, b_off0, b_stride);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// Input causal conv (C_in = 1, src/model.py:84-88,228): y[b][t][c] = b[c] + sum_tap w[tap][c] x[b][t-(KS-1-tap)].
// Arithmetic = the k-ordered fma chain from zero, then + bias: what the fp32 MFMA product it replaces
// computed and what the generation chain kernel (wn_gen.hip) reproduces, so all three agree bit for bit.
__global__ __launch_bounds__(256) void wn_inconv_fwd_kernel(const float* x, const float* w, const float* bias, int B, int T,
                                                            int R, int KS, float* y, float* absmax_out) {
  const int q = R / 4;                                   // float4 groups per row
  const int64_t total = (int64_t)B * T * q;
  float wmax = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    const int64_t row = i / q;
    const int t = (int)(row % T);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int tap = 0; tap < KS; ++tap) {
      const int ts = t - (KS - 1 - tap);
      const float xv = ts >= 0 ? x[row - (KS - 1 - tap)] : 0.f;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)tap * R + c);
      a0 = fmaf(wv.x, xv, a0); a1 = fmaf(wv.y, xv, a1); a2 = fmaf(wv.z, xv, a2); a3 = fmaf(wv.w, xv, a3);
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    const f32x4 o = f32x4{a0 + bv.x, a1 + bv.y, a2 + bv.z, a3 + bv.w};
    *reinterpret_cast<f32x4*>(y + row * R + c) = o;
    wmax = fmaxf(wmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  }
  if (absmax_out) {                                      // forward range guard: running max-abs of the first block input
    // one atomic per WORKGROUP at most: thousands of waves finishing together on one address cost ~15 ns each
    __shared__ float wm[4];
    if (!(wmax < 3.0e38f)) wmax = 3.0e38f;               // inf / NaN count as "beyond any limit"
    wmax = wn_wave_max(wmax);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = wmax;
    __syncthreads();
    if (threadIdx.x == 0) wn_absmax_publish_any(absmax_out, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
  }
}
int wn_launch_inconv_fwd(const float* x, const float* w, const float* bias, int B, int T, int R, int KS, float* y,
                         float* absmax_out, hipStream_t s) {
  const int64_t total = (int64_t)B * T * (R / 4);
  if (total <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_inconv_fwd_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1024)), dim3(256), 0, s, x, w, bias,
                     B, T, R, KS, y, absmax_out);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// Skip path folded into the head's first convolution (training passes).
//   reference: skip_b = W_s(b)^T z_b + b_s(b)  (src/layers.py:216-217), x = sum_b skip_b  (src/model.py:235-236),
//              a = W_f0^T x + b_f0              (first head conv, src/model.py:105-111: conv, THEN activation)
// Both maps are linear with nothing in between, so  a = sum_b V(b)^T z_b + b',  V(b) = W_s(b) W_f0  (D x F0),
// b' = b_f0 + W_f0^T sum_b b_s(b).  With F0 < S every consumer of the S-wide tensors shrinks: the folded forward
// contraction has F0 instead of S output columns, each block's backward reads dL/da (rows x F0) instead of the
// gradient of the skip sum (rows x S), and the weight gradients of all conv_skip AND of the first head conv come from
// ONE rows-contraction  M = Z^T (dL/da)  (N*D x F0):
//   dW_s(b) = M(b) W_f0^T,  db_s(b) = W_f0 colsum(dL/da),  dW_f0 = sum_b W_s(b)^T M(b) + (sum_b b_s) colsum^T,  db_f0 = colsum.
// V and b' are formed below in plain fp32; the small backward products run on the exact-fp32 rows / weight-gradient
// GEMMs (wn_plan.hip) over [M; colsum] and [W_s(all blocks); sum b_s].  The reassociation moves results by ~1e-6
// relative, far inside the 1e-4 parity bar (checked against the oracle, which keeps the reference's order).
// Small dense fp32 product for the folded skip path's weight-space matrices and the conditioning path (a few thousand rows
// at most, operands L2-resident):  C[i][j] = sum_k A[i * sai + k * sak] * B[k * sbk + j * sbj],  i < M, j < N, k < K, C
// row-major with pitch ldc; the generic operand strides cover the plain, the B-transposed and the A-transposed product.
// 32 x 32 tile per workgroup and K in chunks of 128 whose loads are ALL in flight before the first product (one global
// round trip per chunk).  Plain fma chains, k ascending.
// blockIdx.z = batch index: operand / result bases advance by za / zb / zc floats (the conditioning convs of all blocks as
// one launch); bias (+ activation) = the Dense epilogue of the conditioning's mapping stack (src/model.py:121-135).
__global__ __launch_bounds__(256) void wn_sgemm_small32_kernel(const float* A, int64_t sai, int64_t sak, const float* B, int64_t sbk,
                                                               int64_t sbj, float* C, int ldc, int M, int N, int K,
                                                               int64_t za, int64_t zb, int64_t zc, const float* bias, int act,
                                                               int zk) {
  constexpr int KC = 128;
  __shared__ float As[KC][32 + 1], Bs[KC][32 + 1];
  A += (int64_t)blockIdx.z * za; B += (int64_t)blockIdx.z * zb; C += (int64_t)blockIdx.z * zc;
  // zk > 0: split K -- product z covers k in [z zk, min(K, (z + 1) zk)) of ONE long contraction (za / zb advance the
  // operands by zk rows of k); the partial results are summed by the caller
  if (zk > 0) K = max(0, min(K - (int)blockIdx.z * zk, zk));
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // outputs (ty + 8 a, tx), a < 4
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += KC) {
    float ra[16], rb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = threadIdx.x + 256 * q;                           // 32 x 128 elements of each operand
      // the faster-varying index follows the operand's unit stride (coalesced either way round)
      const int ka = sak == 1 ? e % KC : e / 32, ia = sak == 1 ? e / KC : e % 32;
      ra[q] = (i0 + ia < M && k0 + ka < K) ? A[(int64_t)(i0 + ia) * sai + (int64_t)(k0 + ka) * sak] : 0.f;
      const int kb = sbk == 1 ? e % KC : e / 32, jb = sbk == 1 ? e / KC : e % 32;
      rb[q] = (j0 + jb < N && k0 + kb < K) ? B[(int64_t)(k0 + kb) * sbk + (int64_t)(j0 + jb) * sbj] : 0.f;
    }
    if (k0 > 0) __syncthreads();                                     // the previous chunk has been read
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = threadIdx.x + 256 * q;
      const int ka = sak == 1 ? e % KC : e / 32, ia = sak == 1 ? e / KC : e % 32;
      As[ka][ia] = ra[q];
      const int kb = sbk == 1 ? e % KC : e / 32, jb = sbk == 1 ? e / KC : e % 32;
      Bs[kb][jb] = rb[q];
    }
    __syncthreads();
    // (k past the end of K is stored as zero: fma(0, 0, acc) leaves acc as it is; fixed trip count so that the LDS reads
    // of eight k are in flight together)
#pragma unroll 8
    for (int k = 0; k < KC; ++k) {
      const float bv = Bs[k][tx];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = fmaf(As[k][ty + 8 * a], bv, acc[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ty + 8 * a, j = j0 + tx;
    if (i < M && j < N) {
      float v = acc[a];
      if (bias) v = wn_act(v + bias[j], act);
      C[(int64_t)i * ldc + j] = v;
    }
  }
}
int wn_launch_sgemm_small(const float* A, int64_t sai, int64_t sak, const float* B, int64_t sbk, int64_t sbj, float* C, int ldc,
                          int M, int N, int K, hipStream_t s) {
  if (M <= 0 || N <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_sgemm_small32_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, s, A, sai, sak, B, sbk, sbj, C,
                     ldc, M, N, K, (int64_t)0, (int64_t)0, (int64_t)0, (const float*)nullptr, 0, 0);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// nz products in one launch (bases za / zb / zc floats apart), optional Dense epilogue C = act(A B + bias)
int wn_launch_sgemm_small_batched(const float* A, int64_t sai, int64_t sak, int64_t za, const float* B, int64_t sbk, int64_t sbj,
                                  int64_t zb, float* C, int ldc, int64_t zc, int M, int N, int K, int nz, const float* bias,
                                  int act, hipStream_t s, int zk) {
  if (M <= 0 || N <= 0 || nz <= 0) return WN_OK;
  if (nz > 65535) { wn_set_error("sgemm_small_batched: too many products"); return WN_E_UNSUPPORTED; }
  hipLaunchKernelGGL(wn_sgemm_small32_kernel, dim3((N + 31) / 32, (M + 31) / 32, nz), dim3(256), 0, s, A, sai, sak, B, sbk, sbj, C,
                     ldc, M, N, K, za, zb, zc, bias, act, zk);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// gathers every block's W_s into one contiguous [N*D + 1][S] matrix with sum_b b_s as its last row (the blocks' kernels
// are not adjacent in the flat parameter buffer): the A operand of V = W_s(all) W_f0 and of dW_f0 in backward
__global__ void wn_skip_gather_kernel(const float* params, int64_t ws_off0, int64_t ws_stride, const float* bsum, int N, int D,
                                      int S, float* wsall) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(N * D + 1) * S; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / S), sidx = (int)(i % S);
    wsall[i] = k < N * D ? params[ws_off0 + (int64_t)(k / D) * ws_stride + (int64_t)(k % D) * S + sidx] : bsum[sidx];
  }
}
// b' = b_f0 + (last row of V, = W_f0^T sum_b b_s)
__global__ void wn_vec_add_kernel(const float* a, const float* b, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}
int wn_launch_skip_fold(const float* params, int64_t ws_off0, int64_t ws_stride, int64_t wf0_off, int64_t bf0_off,
                        const float* bsum, int N, int D, int S, int F0, float* V, float* bfold, float* wsall, hipStream_t s) {
  const int64_t total = (int64_t)(N * D + 1) * S;
  hipLaunchKernelGGL(wn_skip_gather_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s, params,
                     ws_off0, ws_stride, bsum, N, D, S, wsall);
  WN_HIP_CHECK(hipGetLastError());
  // V ([N*D + 1][F0], the extra row = W_f0^T sum b_s) = wsall W_f0;  W_f0 = kernel (1, S, F0): B[k = s][j = n]
  int rc = wn_launch_sgemm_small(wsall, S, 1, params + wf0_off, F0, 1, V, F0, N * D + 1, F0, S, s);
  if (rc) return rc;
  hipLaunchKernelGGL(wn_vec_add_kernel, dim3((F0 + 255) / 256), dim3(256), 0, s, V + (int64_t)N * D * F0, params + bf0_off, F0, bfold);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// Y = [M; colsum] W_f0^T ([N*D + 1][S], from the rows GEMM): rows b*D.. are dW_s(b), the last row is db_s (the same
// for every block); db_f0 = colsum.  One copy kernel scatters them to the tensors' places in the flat gradient.
__global__ void wn_skip_scatter_kernel(const float* Y, const float* colsum, int64_t ws_off0, int64_t ws_stride, int64_t bs_off0,
                                       int64_t bs_stride, int64_t bf0_off, int N, int D, int S, int F0, float* grads) {
  const int64_t n_ws = (int64_t)N * D * S, n_bs = (int64_t)N * S;
  const int64_t total = n_ws + n_bs + F0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n_ws) {
      const int k = (int)(i / S), sidx = (int)(i % S);
      grads[ws_off0 + (int64_t)(k / D) * ws_stride + (int64_t)(k % D) * S + sidx] = Y[i];
    } else if (i < n_ws + n_bs) {
      const int64_t j = i - n_ws;
      grads[bs_off0 + (j / S) * bs_stride + (j % S)] = Y[n_ws + (j % S)];
    } else {
      grads[bf0_off + (i - n_ws - n_bs)] = colsum[i - n_ws - n_bs];
    }
  }
}
int wn_launch_skip_scatter(const float* Y, const float* colsum, int64_t ws_off0, int64_t ws_stride, int64_t bs_off0,
                           int64_t bs_stride, int64_t bf0_off, int N, int D, int S, int F0, float* grads, hipStream_t s) {
  const int64_t total = (int64_t)N * D * S + (int64_t)N * S + F0;
  hipLaunchKernelGGL(wn_skip_scatter_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, Y, colsum,
                     ws_off0, ws_stride, bs_off0, bs_stride, bf0_off, N, D, S, F0, grads);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

__global__ void wn_guard_flag_kernel(const float* absmax, float limit, int enabled, float* out) {
  const float m = absmax ? *absmax : 0.f;
  out[0] = (enabled && !(m < limit)) ? 1.0f : 0.0f;
}
__global__ void wn_guard_accumulate_kernel(const float* src, float* dst) {
  const unsigned a = __float_as_uint(src[0]) & 0x7fffffffu, b = __float_as_uint(dst[0]) & 0x7fffffffu;
  dst[0] = __uint_as_float(a > b ? a : b);
}
int wn_launch_guard_accumulate(const float* src, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(wn_guard_accumulate_kernel, dim3(1), dim3(1), 0, s, src, dst);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
int wn_launch_guard_flag(const float* absmax, float limit, int enabled, float* out, hipStream_t s) {
  hipLaunchKernelGGL(wn_guard_flag_kernel, dim3(1), dim3(1), 0, s, absmax, limit, enabled, out);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// Weight gradients of the input causal conv (src/model.py:84-88: C_in = 1, KS taps):
//   dW[tap][c] = sum_{b,t} x[b][t - (KS-1-tap)] * g[b][t][c],   db[c] = sum_{b,t} g[b][t][c]
// i.e. KS + 1 weighted column sums of the gradient at the first block input.  On the generic job table the K = 1
// product ran as a handful of single-wave jobs (0.76 ms at configs[1] for 33 MB of operands); here one workgroup
// per (utterance, time range) streams its rows of g once with 16-byte loads -- lanes = 4-channel groups x row
// lanes -- and leaves its partial sums in the split's slab row (flat-gradient layout), like every other weight
// gradient.  Plain fp32 arithmetic.
template <int KS>
__global__ __launch_bounds__(256) void wn_inconv_wgrad_kernel(const float* x, const float* g, int T, int R, int spb,
                                                              float* slab, int64_t P, int64_t w_off, int64_t b_off) {
  __shared__ float red[256 * (KS + 1) * 4];
  const int split = blockIdx.x, ub = split / spb, sp = split % spb;
  int len = (T + spb - 1) / spb;
  const int r0 = sp * len, r1 = min(T, r0 + len);
  const int ng = R / 4, nrl = 256 / ng;                 // 4-channel groups, row lanes
  const int cg = threadIdx.x % ng, rl = threadIdx.x / ng;
  float acc[KS + 1][4];
#pragma unroll
  for (int i = 0; i <= KS; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  const float* xb = x + (int64_t)ub * T;
  const float* gb = g + (int64_t)ub * T * R + 4 * cg;
  for (int t = r0 + rl; t < r1; t += nrl) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gb + (int64_t)t * R);
    const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int tap = 0; tap < KS; ++tap) {
      const int ts = t - (KS - 1 - tap);
      const float xv = ts >= 0 ? xb[ts] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[tap][e] = fmaf(xv, ge[e], acc[tap][e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[KS][e] += ge[e];
  }
#pragma unroll
  for (int i = 0; i <= KS; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(i * 4 + e) * 256 + rl * ng + cg] = acc[i][e];
  __syncthreads();
  // (KS + 1) * R outputs, each the sum over the row lanes
  float* row = slab + (int64_t)split * P;
  for (int o = threadIdx.x; o < (KS + 1) * R; o += 256) {
    const int i = o / R, c = o % R;
    float sum = 0.f;
    for (int l = 0; l < nrl; ++l) sum += red[(i * 4 + (c & 3)) * 256 + l * ng + (c >> 2)];
    if (i < KS) row[w_off + (int64_t)i * R + c] = sum;
    else row[b_off + c] = sum;
  }
}

int wn_inconv_wgrad_supported(int R, int KS) { return (R % 4 == 0 && R >= 4 && R <= 1024 && 256 % (R / 4) == 0 && (KS == 2 || KS == 3)) ? 1 : 0; }

int wn_launch_inconv_wgrad(const float* x, const float* g, int B, int T, int R, int KS, int splits_per_b, float* slab,
                           int64_t P, int64_t w_off, int64_t b_off, hipStream_t s) {
  if (!wn_inconv_wgrad_supported(R, KS)) { wn_set_error("inconv_wgrad: unsupported shape R=%d KS=%d", R, KS); return WN_E_UNSUPPORTED; }
  const dim3 grid((unsigned)(B * splits_per_b));
  if (KS == 2) hipLaunchKernelGGL(wn_inconv_wgrad_kernel<2>, grid, dim3(256), 0, s, x, g, T, R, splits_per_b, slab, P, w_off, b_off);
  else hipLaunchKernelGGL(wn_inconv_wgrad_kernel<3>, grid, dim3(256), 0, s, x, g, T, R, splits_per_b, slab, P, w_off, b_off);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// Queued generation, categorical head, deterministic: softmax -> arg max -> sample value -> output row and
// next network input, in ONE launch.  The arithmetic per row is that of wn_softmax_kernel followed by
// wn_sample_det_cat_kernel (same lane assignment, same reductions), so the result is the same sample.
__global__ __launch_bounds__(256) void wn_gen_tail_cat_det_kernel(const float* logits, int rows, int C, float inv_lv,
                                                                  float* out, int length, int step, float* xin_slot) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float sv = wn_cat_det_row(logits + (int64_t)row * C, C, lane, inv_lv);
  if (lane == 0) {
    out[(int64_t)row * length + step] = sv;
    if (xin_slot) xin_slot[row] = sv;
  }
}
int wn_launch_gen_tail_cat_det(const float* logits, int rows, int C, int bits, float* out, int length, int step,
                               float* xin_slot, hipStream_t s) {
  if (rows <= 0) return WN_OK;
  hipLaunchKernelGGL(wn_gen_tail_cat_det_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, rows, C,
                     1.0f / (float)(1 << (bits - 1)), out, length, step, xin_slot);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
