// Weight gradient of the folded skip path for ALL blocks in one launch (gfx950):
//   dW_s[b][c][n] = sum_rows Z[row][b*D + c] * g_skip[row][n],   db_s[n] = sum_rows g_skip[row][n]
// (src/layers.py:216-217 conv_skip of every block; g_skip is the one gradient of the skip sum,
// src/model.py:236, shared by all blocks).  It is a K = rows GEMM with a 1920 x 256 output at
// configs[1]; the generic job table re-reads g_skip once per 64-column block of Z (30 x 131 MB).
// Here a workgroup of 8 waves owns 256 Z columns x all n (<= 256): wave w keeps its own Z tile
// (A operand) in registers and contributes ONE split g_skip tile to LDS, from which every wave
// reads all n tiles -> g_skip is read ceil(N*D / 256) times, each element is split once per
// workgroup instead of once per job.  Split-precision MFMA (fp16 hi/lo, 3 products), time is the
// MFMA K dimension (16 rows per step).  Partial sums go to the [split][nparams] slab of the batched
// weight-gradient path and are reduced with it.
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

typedef _Float16 ws_h8 __attribute__((ext_vector_type(8)));

struct WnWgSkipArgs {
  const float* z; int32_t ldz;          // block-major [N][rows][D]: ldz = D, plane stride = rows * D
  const float* g; int32_t ldg;          // [rows][S]
  int64_t rows;
  int32_t KZ;                           // N*D
  int32_t S;                            // <= 256, multiple of 32
  int32_t D;                            // columns per block
  int32_t nsplit;
  float* slab; int64_t P;               // slab[split][P]
  int64_t w_off0, w_stride;             // flat offset of block 0's conv_skip kernel, distance between blocks
  int64_t b_off0, b_stride;             // same for the biases
  int32_t nblocks;
  const float* gmax;                    // running max-abs of g_skip (operand scaling) or null
};

// NT = n tiles (S / 32); LDZ / LDG = compile-time row strides of Z and g (0: read from the arguments).
// With static strides the 8 row loads of an operand are one base address + immediate offsets, which
// frees the registers for a second chunk of look-ahead.
template <int NT, int LDZ = 0, int LDG = 0>
__global__ __launch_bounds__(512, 2) void wn_wgrad_skip_kernel(WnWgSkipArgs a) {
  // LDS: two buffers of NT split tiles (hi | lo): NT * 2 KiB each
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 8 * 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 31, h = lane >> 5;
  const int kgroup = blockIdx.y, split = blockIdx.x;
  const int kcol = kgroup * 256 + wave * 32 + tl;          // this lane's Z column (A operand row)
  const bool kok = kcol < a.KZ;
  const bool wave_has_k = kgroup * 256 + wave * 32 < a.KZ;   // wave-uniform
  int64_t len = (a.rows + a.nsplit - 1) / a.nsplit;
  len = (len + 31) & ~(int64_t)31;
  const int64_t r0 = (int64_t)split * len;
  const int64_t r1 = min(a.rows, r0 + len);

  float gsc = 1.0f, inv = 1.0f;
  if (a.gmax) {
    const float m = *a.gmax;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);
      e = max(-100, min(100, e));
      gsc = ldexpf(1.0f, -e);
      inv = ldexpf(1.0f, e);
    }
  }

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float bsum = 0.f;

  // wave w loads Z tile w (its A operand) and g tile (w % NT) ... every g tile must be produced by
  // exactly one wave: waves 0..NT-1 produce g tiles 0..NT-1 (NT <= 8)
  const bool makes_g = wave < NT;
  const int ncol = wave * 32 + tl;                          // g column when makes_g
  const float* zbase = a.z + (int64_t)(kcol / a.D) * a.rows * a.D + (kcol % a.D);   // column kcol -> (block, channel)
  const float* gbase = a.g + ncol;
  auto load = [&](int64_t rr, float (&zv)[8], float (&gv)[8]) {
    const int64_t rb = rr + 8 * h;
    if constexpr (LDZ > 0 && LDG > 0) {
      const float* pz = zbase + rb * LDZ;
      const float* pg = gbase + rb * LDG;
      if (rr + 16 <= r1) {                               // workgroup-uniform: a full chunk needs no row masks
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          zv[e] = kok ? pz[e * LDZ] : 0.f;
          gv[e] = makes_g ? pg[e * LDG] : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          zv[e] = (rb + e < r1 && kok) ? pz[e * LDZ] : 0.f;
          gv[e] = (rb + e < r1 && makes_g) ? pg[e * LDG] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t r = rb + e;
        const bool ok = r < r1;
        zv[e] = (ok && kok) ? zbase[r * a.ldz] : 0.f;
        gv[e] = (ok && makes_g) ? gbase[r * a.ldg] : 0.f;
      }
    }
  };
  auto split8 = [&](const float (&v)[8], float s, ws_h8& hi, ws_h8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 hh = (_Float16)(v[e] * s);
      hi[e] = hh;
      lo[e] = (_Float16)__builtin_fmaf(v[e], s, -(float)hh);     // product unrounded (explicit fma)
    }
  };

  int buf = 0;
  auto compute = [&](const float (&zv)[8], const float (&gv)[8]) {
    ws_h8 zh, zl, gh, gl;
    split8(zv, 1.0f, zh, zl);
    if (makes_g) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum += gv[e];
      split8(gv, gsc, gh, gl);
      ws_h8* dst = reinterpret_cast<ws_h8*>(smem + buf * (8 * 2048) + wave * 2048);
      dst[lane] = gh;
      dst[64 + lane] = gl;
    }
    __syncthreads();
    const ws_h8* src = reinterpret_cast<const ws_h8*>(smem + buf * (8 * 2048));
    if (wave_has_k) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const ws_h8 bh = src[j * 128 + lane], bl = src[j * 128 + 64 + lane];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zl, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, bh, acc[j], 0, 0, 0);
      }
    }
    buf ^= 1;
  };
  if constexpr (LDZ > 0 && LDG > 0) {
    // three register sets: the raw operands of the next TWO chunks are in flight during a chunk's MFMAs
    float zr[3][8], gr[3][8];
    int64_t rr = r0;
    if (rr < r1) load(rr, zr[0], gr[0]);
    if (rr + 16 < r1) load(rr + 16, zr[1], gr[1]);
    for (; rr < r1; rr += 48) {
      if (rr + 32 < r1) load(rr + 32, zr[2], gr[2]);
      compute(zr[0], gr[0]);
      if (rr + 16 >= r1) break;
      if (rr + 48 < r1) load(rr + 48, zr[0], gr[0]);
      compute(zr[1], gr[1]);
      if (rr + 32 >= r1) break;
      if (rr + 64 < r1) load(rr + 64, zr[1], gr[1]);
      compute(zr[2], gr[2]);
    }
  } else {
    float z0[8], g0[8], z1[8], g1[8];
    int64_t rr = r0;
    if (rr < r1) load(rr, z0, g0);
    for (; rr < r1; rr += 32) {
      if (rr + 16 < r1) load(rr + 16, z1, g1);
      compute(z0, g0);
      if (rr + 16 >= r1) break;
      if (rr + 32 < r1) load(rr + 32, z0, g0);
      compute(z1, g1);
    }
  }

  // ---- store: dW_s of block b = k / D, row c = k % D ----
  float* row = a.slab + (int64_t)split * a.P;
  if (wave_has_k) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kgroup * 256 + wave * 32 + wn_drow(r, h);
        const int n = 32 * j + tl;
        if (k < a.KZ) row[a.w_off0 + (int64_t)(k / a.D) * a.w_stride + (int64_t)(k % a.D) * a.S + n] = acc[j][r] * inv;
      }
  }
  if (kgroup == 0 && makes_g) {
    const float tot = bsum + __shfl_xor(bsum, 32);
    if (h == 0)
      for (int b = 0; b < a.nblocks; ++b) row[a.b_off0 + (int64_t)b * a.b_stride + ncol] = tot;
  }
}

int wn_wgrad_skip_supported(int D, int S, int KZ) {
  return (S % 32 == 0 && S >= 32 && S <= 256 && D % 32 == 0 && KZ % 32 == 0) ? 1 : 0;
}

int wn_launch_wgrad_skip(const float* z, int ldz, const float* g, int ldg, int64_t rows, int KZ, int S, int D,
                         int nsplit, float* slab, int64_t P, int64_t w_off0, int64_t w_stride, int64_t b_off0,
                         int64_t b_stride, int nblocks, const float* gmax, hipStream_t s) {
  WnWgSkipArgs a;
  a.z = z; a.ldz = ldz; a.g = g; a.ldg = ldg; a.rows = rows; a.KZ = KZ; a.S = S; a.D = D; a.nsplit = nsplit;
  a.slab = slab; a.P = P; a.w_off0 = w_off0; a.w_stride = w_stride; a.b_off0 = b_off0; a.b_stride = b_stride;
  a.nblocks = nblocks; a.gmax = gmax;
  dim3 grid(nsplit, (KZ + 255) / 256);
  switch (S / 32) {
    case 1: hipLaunchKernelGGL(wn_wgrad_skip_kernel<1>, grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL(wn_wgrad_skip_kernel<2>, grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL(wn_wgrad_skip_kernel<3>, grid, dim3(512), 0, s, a); break;
    case 4:
      if (ldz == 64 && ldg == 128) hipLaunchKernelGGL((wn_wgrad_skip_kernel<4, 64, 128>), grid, dim3(512), 0, s, a);
      else if (ldz == 128 && ldg == 128) hipLaunchKernelGGL((wn_wgrad_skip_kernel<4, 128, 128>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL(wn_wgrad_skip_kernel<4>, grid, dim3(512), 0, s, a);
      break;
    case 5: hipLaunchKernelGGL(wn_wgrad_skip_kernel<5>, grid, dim3(512), 0, s, a); break;
    case 6: hipLaunchKernelGGL(wn_wgrad_skip_kernel<6>, grid, dim3(512), 0, s, a); break;
    case 7: hipLaunchKernelGGL(wn_wgrad_skip_kernel<7>, grid, dim3(512), 0, s, a); break;
    default:
      if (ldz == 64 && ldg == 256) hipLaunchKernelGGL((wn_wgrad_skip_kernel<8, 64, 256>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL(wn_wgrad_skip_kernel<8>, grid, dim3(512), 0, s, a);
      break;
  }
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
