// Building blocks of the streamed-weights kernels (gfx950): LDS-DMA by inline assembly, the split of an fp32 operand into
// fp16 hi | lo, and the store of a 32 x 32 accumulator tile as whole row segments.  (wn_layer16s.hip and wn_bwd16s.hip carry
// their own, older copies of the first and the last.)
#pragma once
#include <hip/hip_fp16.h>

#include "wn_kernels.h"

namespace wn_stream {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// hi = fp16(q), lo = fp16(q - hi): the forward operand split of wn_gemm16.hip (scale 1)
__device__ __forceinline__ void split8(const f32x4& q0, const f32x4& q1, h8& hi, h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)v[e];
    hi[e] = h;
    lo[e] = (_Float16)(v[e] - (float)h);
  }
}

// the same with the operand scaled by s (an exact power of two): hi = fp16(q s), lo = fp16(q s - hi) with the product
// unrounded -- wn_split8g of wn_gemm16.hip
__device__ __forceinline__ void split8s(const f32x4& q0, const f32x4& q1, float s, h8& hi, h8& lo) {
  const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)(v[e] * s);
    hi[e] = h;
    lo[e] = (_Float16)__builtin_fmaf(v[e], s, -(float)h);
  }
}

// LDS-DMA of 16 bytes per lane: global address = scalar base + 32-bit lane offset, LDS address = M0 + lane * 16.
// Inline assembly on purpose: through the builtin hipcc forms every address as a 64-bit VGPR pair, hoists the pairs out of
// the loop, spills them and reloads each with s_waitcnt vmcnt(0) in front of its request.  The
// compiler does not count these requests in its own vmcnt bookkeeping: its waits only become more conservative.
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");   // (m0 is reserved: the compiler never keeps a value in it)
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

// one 32 x 32 D-layout accumulator tile -> wave-private LDS stage (32 rows of PITCH floats) -> 128-byte row segments in
// HBM.  dst = wave-uniform address of the tile's first row (+ column offset), voff = this lane's byte offset inside a
// group of eight rows.  The base goes through an empty asm so that it stays ONE scalar (otherwise hipcc hoists a 64-bit
// VGPR pair per output tensor out of the tile loop and spills it); the asm drops the address space, which is restored, or
// the stores become flat_store.  FULL: all 32 rows exist.
template <int PITCH, bool FULL>
__device__ __forceinline__ void store_tile(const f32x16& v, float* stage, float* dst, unsigned voff, unsigned ld_bytes,
                                           int rows_valid, int lane) {
  const int tl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 o;
    o.x = v[4 * rq + 0]; o.y = v[4 * rq + 1]; o.z = v[4 * rq + 2]; o.w = v[4 * rq + 3];
    *reinterpret_cast<f32x4*>(stage + tl * PITCH + 8 * rq + 4 * h) = o;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float* rd = stage + (lane >> 3) * PITCH + (lane & 7) * 4;
  char* base0 = reinterpret_cast<char*>(dst);
  asm volatile("" : "+s"(base0));
  __attribute__((address_space(1))) char* base = (__attribute__((address_space(1))) char*)base0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(rd + i * 8 * PITCH);
    if (FULL || i * 8 + (lane >> 3) < rows_valid)
      *(__attribute__((address_space(1))) f32x4*)(base + (uint64_t)((unsigned)(i * 8) * ld_bytes) + voff) = o;
  }
  asm volatile("" ::: "memory");
}

}  // namespace wn_stream
