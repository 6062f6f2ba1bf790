// Rows contraction over a PLANAR operand with 128 or 256 output columns, streamed weights, second form (gfx950):
//   Y[t][n] = act( b[n] + sum_p sum_k Z_p[t][k] * W[p * PK + k][n] ),   n < N, p < NP planes of PK channels each.
// Users: the head's 1x1 convs with 128 / 256 outputs (one plane; src/model.py:105-119) and the folded skip contraction: the skip sum over all blocks' gated activations
// (src/layers.py:216-217, src/model.py:235-236) folded with the head's first conv (src/model.py:105-111): K = N_blocks * D.
//
// Same arithmetic as wn_gemm_rows16_kernel<4, PLAIN> on the same fp16 hi|lo image -- accumulators from zero, k ascending,
// lo*hi, hi*lo, hi*hi per k-step, then + bias, activation -- so the two kernels agree bit for bit (tested).  What differs is
// the pipeline, taken from wn_layer16s.hip:
//   * workgroups of FOUR waves, two per CU, instead of one of eight: one workgroup's barrier waits and tile stores overlap
//     the other's products;
//   * the weight stream (8 KiB per k-step: 4 column tiles x hi|lo) runs through a 3-deep LDS ring, two k-steps ahead, and
//     never stops at a tile boundary (the old kernel drained and refilled its ring per tile);
//   * a wave owns RT = 2 row tiles (64 time steps): every weight fragment read from LDS feeds two products, and the image
//     is streamed from L2 once per 256 rows instead of once per 128 (the old kernel's variant of this spilled);
//   * every request is an inline-assembly LDS-DMA from a scalar base (wn_stream.h), the only waits are counted vmcnt.
// Measured: configs[1] (K = 1920) 294 -> 270 us, configs[3] (K = 3840) 0.73 -> 0.71 ms: the pipeline was not what held this
// contraction back (two k-steps of operands per wave in flight are right at bandwidth x latency).
#include "wn_stream.h"
#include "wn_sample.h"

using namespace wn_stream;

namespace {

template <int RT, int JT_>
struct GS {
  static constexpr int JT = JT_;                       // column tiles: 4 (128 outputs, two row tiles per wave) or 8 (256, one)
  static constexpr int CHUNK = JT * 2048, NBUF = 3;    // one k-step of weights
  static constexpr int XB = 3, XBUF = RT * 2048;       // a wave's activations of one k-step
  static constexpr int WAVES = 4, THREADS = 256;
  static constexpr int PITCH = 36, STAGE = 32 * PITCH * 4;
  static constexpr int REGION = XB * XBUF;             // per wave: activation ring, reused as the output stage
  static constexpr int LDS = NBUF * CHUNK + WAVES * REGION + 32 * JT * 4;    // + bias table
  static constexpr int PT = CHUNK / 16 / THREADS;      // weight requests per thread and k-step (2)
  static constexpr int PX = 2 * RT;                    // activation requests per lane and k-step
  static_assert(STAGE <= REGION, "the output stage lives in the activation ring");
};

}  // namespace

// ACT >= 0: the activation fixed at compile time (linear / relu / leaky relu: straight-line epilogue); -1: a.act at run time;
// -2: the backward-data form -- the operand is a gradient, scaled by an exact power of two from its running max-abs
// (a.absmax_in) like wn_gemm_rows16_kernel does, the epilogue multiplies by act'(saved output) (a.aux) instead of applying
// act, the result's max-abs goes to a gradient slot
// SHIFT: the planes are the taps of a dilated conv over ONE tensor (plane_stride 0): plane p reads row t - a.shift[p] of the
// same utterance, rows outside [0, T) contribute zero (src/layers.py:66-88 causal padding; negative shifts: its backward).
// JT = 2 serves 32 and 64 output channels (a.N; a 32-channel image is padded to two row tiles, see wn_gemm_rows16_ok).
// -3: the categorical head's last conv with the loss as its epilogue (RT = 1, JT = 8: 256 classes): a lane holds HALF a
// row of logits in its 128 accumulator registers -- class 32 j + 8 (r >> 2) + 4 h + (r & 3) in register r of tile j, the
// other half in lane ^ 32 -- so softmax, the clipped cross entropy of src/model.py:515-516, its gradient and the
// sample_waveform draw are register loops plus a handful of exchanges with the partner lane, for 32 rows at once; the
// one-row-per-wave loss kernel spends ~300 wave instructions per ROW on the same arithmetic (wn_cat_loss256_kernel: VALU
// bound at 105 us for 268 MB) and needs the logits written (131 MB) and read back (131 MB).
template <int RT, int JT_, int ACT, bool SHIFT = false>
__global__ __launch_bounds__(256, 2) void wn_gemm_planes16s_kernel(WnGemmPlanesArgs a) {
  using C = GS<RT, JT_>;
  constexpr int JT = C::JT, PT = C::PT, PX = C::PX, PITCH = C::PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 31, h = lane >> 5;
  unsigned char* const xbuf = smem + C::NBUF * C::CHUNK + wave * C::REGION;
  float* stage = reinterpret_cast<float*>(xbuf);
  const unsigned smem_addr = lds_addr_of(smem), xbuf_addr = lds_addr_of(xbuf);
  float* sbias = reinterpret_cast<float*>(smem + C::NBUF * C::CHUNK + C::WAVES * C::REGION);
  if (tid < 32 * JT) sbias[tid] = (a.bias && tid < a.N) ? a.bias[tid] : 0.f;
  const bool has_bias = a.bias != nullptr;
  __syncthreads();
  float sc = 1.0f, inv_sc = 1.0f;
  if (ACT == -2 && a.absmax_in) {
    const float m = *a.absmax_in;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);               // m = f * 2^e, f in [0.5, 1)
      e = max(-100, min(100, e));
      sc = ldexpf(1.0f, -e);             // scaled values lie in [-1, 1)
      inv_sc = ldexpf(1.0f, e);
    }
  }

  const int kpp = a.plane_k >> 4;                      // k-steps per plane
  const int nsteps = a.nplanes * kpp;                  // >= 3 (launcher)
  const int tiles_per_b = (a.T + 32 * RT - 1) / (32 * RT);
  const int64_t ntiles = (int64_t)a.B * tiles_per_b;
  const int64_t per_pass = (int64_t)gridDim.x * C::WAVES;
  const int passes = (int)((ntiles + per_pass - 1) / per_pass);

  // the weight stream: k-step ws_next of the image goes to ring slot ws_slot; it wraps at nsteps (next tile)
  const unsigned woff = (unsigned)tid * 16u;
  const char* const wimg = reinterpret_cast<const char*>(a.w16);
  int ws_next = 0, ws_slot = 0;                        // scalar
  auto wpiece = [&](int i) {
    dma16(wimg + (int64_t)ws_next * C::CHUNK + 4096 * i, woff, smem_addr + ws_slot * C::CHUNK + (C::THREADS * i + wave * 64) * 16);
  };
  auto wadvance = [&]() {
    ws_next = ws_next + 1 == nsteps ? 0 : ws_next + 1;
    ws_slot = ws_slot + 1 == C::NBUF ? 0 : ws_slot + 1;
  };
#pragma unroll
  for (int i = 0; i < PT; ++i) wpiece(i);
  wadvance();
#pragma unroll
  for (int i = 0; i < PT; ++i) wpiece(i);
  wadvance();
  int slot = 0;                                        // ring slot of the current k-step's weights (scalar)
  float wmax = 0.f;

  for (int pass = 0; pass < passes; ++pass) {
    const int64_t tile = ((int64_t)pass * gridDim.x + blockIdx.x) * C::WAVES + wave;
    const bool live = tile < ntiles;                   // dead waves still take part in the barriers and the weight stream
    const int b = live ? (int)(tile / tiles_per_b) : 0;
    const int t0 = live ? (int)(tile % tiles_per_b) * 32 * RT : 0;
    const int64_t row0 = (int64_t)b * a.T + t0;
    unsigned xoff[RT];
    bool xok[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int t = t0 + 32 * rt + tl;
      xok[rt] = live && t < a.T;
      xoff[rt] = (unsigned)(((int64_t)b * a.T + (xok[rt] ? t : 0)) * a.ld + 4 * h) * 4u;   // (launcher: a plane stays below 4 GiB)
    }
    // the activation stream of this tile: k-step xs_k of plane base xs_base goes to activation buffer xs_buf
    const char* xs_base = reinterpret_cast<const char*>(a.z);
    int xs_k = 0, xs_buf = 0, xs_p = 0;                // scalar
    // SHIFT: this lane's row of plane p (clamped; rows outside the utterance are zeroed at use)
    auto shifted = [&](int rt, int pl, bool& ok) -> unsigned {
      // (selects, not an indexed read: a runtime index would move the argument array to scratch)
      const int sh = pl == 0 ? a.shift[0] : (pl == 1 ? a.shift[1] : (pl == 2 ? a.shift[2] : a.shift[3]));
      const int t = t0 + 32 * rt + tl, ts = t - sh;
      ok = live && t < a.T && ts >= 0 && ts < a.T;
      return (unsigned)(((int64_t)b * a.T + (ok ? ts : 0)) * a.ld + 4 * h) * 4u;
    };
    auto xdma = [&]() {
      const char* base = xs_base + 64 * xs_k;
      const unsigned dst = xbuf_addr + xs_buf * C::XBUF;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bool okd;
        const unsigned off = SHIFT ? shifted(rt, xs_p, okd) : xoff[rt];
        dma16(base, off, dst + rt * 2048);
        dma16(base + 32, off, dst + rt * 2048 + 1024);
      }
      if (++xs_k == kpp) { xs_k = 0; xs_base += a.plane_stride * 4; ++xs_p; }
      xs_buf = xs_buf + 1 == C::XB ? 0 : xs_buf + 1;
    };

    f32x16 acc[RT][JT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][j][r] = 0.f;

    // Issue order per tile:  x0 x1 | step c: wait, barrier, then w(c+2) x(c+2) between the products of k-step c.  (w(0), w(1)
    // of THIS tile were requested during the previous tile's last two steps, or before the loop.)  vmcnt retires loads in
    // order: "at most N outstanding" with N = the loads issued after the ones step c needs means those have landed.
    xdma();
    xdma();
    int xcur = 0;                                      // activation buffer of the current step (scalar)
    int cs_k = 0, cs_p = 0;                            // k-step inside the plane and plane of the current step (SHIFT)
    // WAIT = loads younger than x(c); XNEXT: x(c + 2) exists
    auto step = [&](auto wait_, bool xnext) {
      constexpr int WAIT = decltype(wait_)::value;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
      asm volatile("s_barrier" ::: "memory");
      const h8* wl = reinterpret_cast<const h8*>(smem + slot * C::CHUNK) + lane;
      const f32x4* xl = reinterpret_cast<const f32x4*>(xbuf + xcur * C::XBUF) + lane;
      h8 bh[RT], bl[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 q0 = xl[rt * 128], q1 = xl[rt * 128 + 64];
        bool okc = xok[rt];
        if constexpr (SHIFT) (void)shifted(rt, cs_p, okc);
        if (!okc) { q0 = f32x4{0.f, 0.f, 0.f, 0.f}; q1 = q0; }
        if constexpr (ACT == -2) split8s(q0, q1, sc, bh[rt], bl[rt]);
        else split8(q0, q1, bh[rt], bl[rt]);
      }
      h8 fr[2][2];
      fr[0][0] = wl[0];
      fr[0][1] = wl[64];
      wn_static_for<JT>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j + 1 < JT) {
          fr[(j + 1) & 1][0] = wl[((j + 1) * 2 + 0) * 64];
          fr[(j + 1) & 1][1] = wl[((j + 1) * 2 + 1) * 64];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[rt][j] = mfma16(fr[j & 1][1], bh[rt], acc[rt][j]);
          acc[rt][j] = mfma16(fr[j & 1][0], bl[rt], acc[rt][j]);
          acc[rt][j] = mfma16(fr[j & 1][0], bh[rt], acc[rt][j]);
        }
        // one request group of the look-ahead k-step per product block
        if constexpr (j < PT) wpiece(j);
        if constexpr (j == PT) { if (xnext) xdma(); }
        __builtin_amdgcn_sched_barrier(0);
      });
      wadvance();
      slot = slot + 1 == C::NBUF ? 0 : slot + 1;
      xcur = xcur + 1 == C::XB ? 0 : xcur + 1;
      if (++cs_k == kpp) { cs_k = 0; ++cs_p; }
    };
    step(std::integral_constant<int, PX>{}, true);                                   // c = 0: younger = x(1)
    for (int c = 1; c + 1 < nsteps; ++c) step(std::integral_constant<int, PT + PX>{}, c + 2 < nsteps);
    step(std::integral_constant<int, PT>{}, false);                                  // c = nsteps - 1: younger = w(c + 1)

    // ---- bias, activation, range guard, staged row stores ----
    const unsigned voff = (unsigned)(lane >> 3) * (unsigned)(a.ldy * 4) + (unsigned)(lane & 7) * 16u;
    if constexpr (ACT == -3) {
      static_assert(ACT != -3 || (RT == 1 && JT == 8), "the loss epilogue is the 256-class form");
      const int rows_valid = live ? max(0, min(32, a.T - t0)) : 0;
      const bool rok = xok[0];
      const int64_t row = row0 + tl;
      int tgt = rok ? a.target[row] : 0;
      tgt = tgt < 0 ? 0 : (tgt > 255 ? 255 : tgt);
      auto xch = [](float v) { return __shfl_xor(v, 32); };       // the partner lane holds the row's other 128 classes
      // logits = acc + bias; row maximum
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + 32 * j + 8 * rq + 4 * h);
          acc[0][j][4 * rq + 0] += bv.x; acc[0][j][4 * rq + 1] += bv.y; acc[0][j][4 * rq + 2] += bv.z; acc[0][j][4 * rq + 3] += bv.w;
          m = fmaxf(fmaxf(m, fmaxf(acc[0][j][4 * rq + 0], acc[0][j][4 * rq + 1])), fmaxf(acc[0][j][4 * rq + 2], acc[0][j][4 * rq + 3]));
        }
      m = fmaxf(m, xch(m));
      // e = exp(l - m) in place, per column tile (32 contiguous classes) so that the draw below can reuse the tile sums
      float tsum[JT];
      float z = 0.f;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        float ts = 0.f;
#pragma unroll
        // (v_exp_f32 on the scaled argument, as the gate's sigmoid: |relative error| ~ 1e-7, arguments <= 0)
        for (int r = 0; r < 16; ++r) { acc[0][j][r] = __builtin_amdgcn_exp2f(1.4426950408889634f * (acc[0][j][r] - m)); ts += acc[0][j][r]; }
        ts += xch(ts);                                    // (commutative: both lanes of a row hold the same tile sum)
        tsum[j] = ts;
        z += ts;
      }
      const float inv = 1.0f / z;
      // ---- sample_waveform(pred) of the row: inverse CDF over the classes in order (src/model.py:405-411), the draw
      // wn_draw_cat_row makes from these probabilities up to the rounding of its running sums ----
      if (a.sample_out) {
        uint32_t rnd[4];
        wn_philox((uint64_t)row, a.offset, a.seed, rnd);
        const float thr = wn_u01(rnd[0]) * (z * inv);     // target mass in probability units (total = sum of q)
        // column tile that holds the crossing
        float run = 0.f, before = 0.f;
        int jsel = JT - 1;
        bool found = false;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const float nxt = run + tsum[j] * inv;
          if (!found && nxt > thr) { found = true; jsel = j; before = run; }
          run = nxt;
        }
        // this lane's 16 probabilities of that tile (a select chain: the tile index differs from row to row)
        float qv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[0][0][r];
#pragma unroll
          for (int j = 1; j < JT; ++j) v = jsel == j ? acc[0][j][r] : v;
          qv[r] = v * inv;
        }
        // groups of four contiguous classes in class order: k = 2 rq + h; this lane owns the groups with its h and gets
        // the partner's sums.  Both lanes walk the same sums in the same order, so they agree on the crossing group.
        float gown[4], goth[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          gown[rq] = (qv[4 * rq] + qv[4 * rq + 1]) + (qv[4 * rq + 2] + qv[4 * rq + 3]);
          goth[rq] = xch(gown[rq]);
        }
        int kstar = 7;
        float runb = before, rr = before;
        bool fk = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gk = ((k & 1) == h) ? gown[k >> 1] : goth[k >> 1];
          const float nxt = rr + gk;
          if (!fk && nxt > thr) { fk = true; kstar = k; runb = rr; }
          rr = nxt;
        }
        const int rqs = kstar >> 1;
        float r_ = runb;
        int es = 3;
        bool fe = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ve = rqs == 0 ? qv[e] : (rqs == 1 ? qv[4 + e] : (rqs == 2 ? qv[8 + e] : qv[12 + e]));
          r_ += ve;
          if (!fe && r_ > thr) { fe = true; es = e; }
        }
        const int mine = 32 * jsel + 8 * rqs + 4 * h + es;       // meaningful in the lane that owns group kstar
        const int theirs = __shfl_xor(mine, 32);
        const int drawn = ((kstar & 1) == h) ? mine : theirs;
        if (rok && h == 0) a.sample_out[row] = (float)drawn * a.inv_lv - 1.0f;
      }
      // ---- clipped probabilities: S = sum clip(q), A = sum of the q inside the clip range, the target's q ----
      float S = 0.f, A = 0.f, qt = 0.f;
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float q = acc[0][j][r] * inv;
          acc[0][j][r] = q;
          S += fminf(fmaxf(q, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
          if (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) A += q;
          if (32 * j + 8 * (r >> 2) + 4 * h + (r & 3) == tgt) qt = q;
        }
      S += xch(S);
      A += xch(A);
      qt += xch(qt);                                      // (the other half contributes 0)
      const float pt = fminf(fmaxf(qt, WN_KERAS_EPS), 1.0f - WN_KERAS_EPS);
      const float ct = (qt >= WN_KERAS_EPS && qt <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
      if (rok && h == 0) a.loss_rows[row] = -(logf(pt) - logf(S));
      // ---- gradient w.r.t. the logits, in place ----
      const float invS = 1.0f / S;
      const float dot = A * invS - ct * qt / pt;          // sum_j g_j q_j
#pragma unroll
      for (int j = 0; j < JT; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float q = acc[0][j][r];
          const float c = (q >= WN_KERAS_EPS && q <= 1.0f - WN_KERAS_EPS) ? 1.f : 0.f;
          float g = c * invS;
          if (32 * j + 8 * (r >> 2) + 4 * h + (r & 3) == tgt) g -= ct / pt;
          const float gl = a.gscale * q * (g - dot);
          acc[0][j][r] = gl;
          if (rok) wmax = fmaxf(wmax, fabsf(gl));
        }
        if (rows_valid > 0) {
          float* dst = a.y + row0 * a.ldy + 32 * j;
          if (rows_valid == 32) store_tile<PITCH, true>(acc[0][j], stage, dst, voff, (unsigned)(a.ldy * 4), rows_valid, lane);
          else store_tile<PITCH, false>(acc[0][j], stage, dst, voff, (unsigned)(a.ldy * 4), rows_valid, lane);
        }
      }
      continue;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int rows_valid = live ? max(0, min(32, a.T - (t0 + 32 * rt))) : 0;
      if (rows_valid <= 0) continue;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        if (32 * j >= a.N) continue;                     // (padded image: the second row tile of a 32-channel output)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int n0 = 32 * j + 8 * rq + 4 * h;
          float v[4] = {acc[rt][j][4 * rq + 0], acc[rt][j][4 * rq + 1], acc[rt][j][4 * rq + 2], acc[rt][j][4 * rq + 3]};
          if constexpr (ACT == -2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= inv_sc;
          }
          if (has_bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + n0);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if constexpr (ACT == -2) {
            if (a.aux && xok[rt]) {
              const f32x4 yv = *(const __attribute__((address_space(1))) f32x4*)(a.aux + (row0 + 32 * rt + tl) * a.ld_aux + n0);
              v[0] *= wn_dact_from_y(yv.x, a.act); v[1] *= wn_dact_from_y(yv.y, a.act);
              v[2] *= wn_dact_from_y(yv.z, a.act); v[3] *= wn_dact_from_y(yv.w, a.act);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = wn_act(v[e], ACT >= 0 ? ACT : a.act);
          }
          if (xok[rt]) wmax = wn_absmax_acc(wmax, v[0], v[1], v[2], v[3]);
          acc[rt][j][4 * rq + 0] = v[0]; acc[rt][j][4 * rq + 1] = v[1]; acc[rt][j][4 * rq + 2] = v[2]; acc[rt][j][4 * rq + 3] = v[3];
        }
        float* dst = a.y + (row0 + 32 * rt) * a.ldy + 32 * j;
        if (rows_valid == 32) store_tile<PITCH, true>(acc[rt][j], stage, dst, voff, (unsigned)(a.ldy * 4), rows_valid, lane);
        else store_tile<PITCH, false>(acc[rt][j], stage, dst, voff, (unsigned)(a.ldy * 4), rows_valid, lane);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead k-steps land before the LDS is given back
  if (a.absmax_out) {
    wmax = wn_wave_absmax_bits(wmax);
    if (lane == 0) { if (ACT == -2 || ACT == -3) wn_absmax_publish(a.absmax_out, wmax); else wn_absmax_publish_any(a.absmax_out, wmax); }
  }
}

// the shifted-plane form: 32 / 64 output channels, the planes are <= 4 taps of one tensor
int wn_gemm_taps16s_supported(int N, int plane_k, int ntaps, int ld, int ldy) {
  return (N == 32 || N == 64) && ntaps >= 1 && ntaps <= 4 && plane_k >= 16 && plane_k % 16 == 0 && ntaps * (plane_k / 16) >= 3 &&
         ld % 4 == 0 && ld >= plane_k && ldy % 4 == 0 && ldy >= N;
}

int wn_gemm_planes16s_supported(int N, int plane_k, int nplanes, int ld, int ldy) {
  return (N == 128 || N == 256) && plane_k >= 16 && plane_k % 16 == 0 && nplanes * (plane_k / 16) >= 3 && ld % 4 == 0 && ld >= plane_k && ldy % 4 == 0 &&
         ldy >= N;
}

int wn_launch_gemm_planes16s(const WnGemmPlanesArgs& a, hipStream_t s) {
  if (!(a.nshift > 0 ? wn_gemm_taps16s_supported(a.N, a.plane_k, a.nplanes, a.ld, a.ldy)
                     : wn_gemm_planes16s_supported(a.N, a.plane_k, a.nplanes, a.ld, a.ldy))) {
    wn_set_error("gemm_planes16s: unsupported shape N=%d plane_k=%d planes=%d", a.N, a.plane_k, a.nplanes);
    return WN_E_UNSUPPORTED;
  }
  if ((int64_t)a.B * a.T * a.ld * 4 >= (int64_t)1 << 32) { wn_set_error("gemm_planes16s: plane beyond 4 GiB"); return WN_E_UNSUPPORTED; }
  if ((int64_t)a.B * a.T <= 0) return WN_OK;
  // 128 columns: two row tiles per wave; 256 columns: one (128 accumulator registers either way)
  const int rt = a.N == 256 ? 1 : 2;
  const int64_t tiles = (int64_t)a.B * ((a.T + 32 * rt - 1) / (32 * rt));
  int64_t gx = (tiles + 3) / 4;
  if (gx > 512) gx = 512;                                // two persistent workgroups of four waves per CU
#define WN_GS_LAUNCH(RT_, JT_, ACT_) hipLaunchKernelGGL((wn_gemm_planes16s_kernel<RT_, JT_, ACT_>), dim3((unsigned)gx), dim3(256), 0, s, a)
#define WN_GS_LAUNCH_S(ACT_) hipLaunchKernelGGL((wn_gemm_planes16s_kernel<2, 2, ACT_, true>), dim3((unsigned)gx), dim3(256), 0, s, a)
  if (a.N <= 64) {
    // the convs of a stack deeper than 1 (32 / 64 output channels, taps as shifted planes of one tensor)
    if (a.nshift != a.nplanes || a.plane_stride != 0 || a.nplanes > 4) { wn_set_error("gemm_planes16s: narrow outputs are the shifted form"); return WN_E_UNSUPPORTED; }
    if (a.bwd) WN_GS_LAUNCH_S(-2);
    else switch (a.act) {
      case WN_ACT_LINEAR: WN_GS_LAUNCH_S(WN_ACT_LINEAR); break;
      case WN_ACT_RELU: WN_GS_LAUNCH_S(WN_ACT_RELU); break;
      case WN_ACT_LEAKY_RELU: WN_GS_LAUNCH_S(WN_ACT_LEAKY_RELU); break;
      default: WN_GS_LAUNCH_S(-1); break;
    }
  } else if (a.cat_loss) {
    if (a.N != 256 || a.bwd || !a.target || !a.loss_rows || !a.y) { wn_set_error("gemm_planes16s: the loss epilogue is the 256-class forward form"); return WN_E_UNSUPPORTED; }
    WN_GS_LAUNCH(1, 8, -3);
  } else if (a.bwd) {
    if (a.N == 256) WN_GS_LAUNCH(1, 8, -2);
    else WN_GS_LAUNCH(2, 4, -2);
  } else if (a.N == 256) switch (a.act) {
    case WN_ACT_LINEAR: WN_GS_LAUNCH(1, 8, WN_ACT_LINEAR); break;
    case WN_ACT_RELU: WN_GS_LAUNCH(1, 8, WN_ACT_RELU); break;
    case WN_ACT_LEAKY_RELU: WN_GS_LAUNCH(1, 8, WN_ACT_LEAKY_RELU); break;
    default: WN_GS_LAUNCH(1, 8, -1); break;
  }
  else switch (a.act) {
    case WN_ACT_LINEAR: WN_GS_LAUNCH(2, 4, WN_ACT_LINEAR); break;
    case WN_ACT_RELU: WN_GS_LAUNCH(2, 4, WN_ACT_RELU); break;
    case WN_ACT_LEAKY_RELU: WN_GS_LAUNCH(2, 4, WN_ACT_LEAKY_RELU); break;
    default: WN_GS_LAUNCH(2, 4, -1); break;
  }
#undef WN_GS_LAUNCH
#undef WN_GS_LAUNCH_S
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}
