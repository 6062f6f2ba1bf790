// The model's forward pass and loss: WaveNet.call src/model.py:213-239, loss_fn :505-551, test_step's loss :362-381.
#include "wn_plan_internal.h"

namespace {

__global__ void wn_ring_capture_kernel(const float* src, int B, int T, int C, int nslots, float* ring) {
  // ring[(t % nslots)][b][c] = src[b][t][c] for the last nslots time steps
  const int64_t n = (int64_t)nslots * B * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int b = (int)((i / C) % B);
    const int k = (int)(i / ((int64_t)C * B));
    const int t = T - nslots + k;
    if (t >= 0) ring[((int64_t)(t % nslots) * B + b) * C + c] = src[((int64_t)b * T + t) * C + c];
  }
}

__global__ void wn_shift_split_kernel(const float* x_full, int B, int T, float* inputs, float* y_true) {
  const int64_t n = (int64_t)B * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / T, t = i % T;
    inputs[i] = x_full[b * (T + 1) + t];        // x[:, :-1]   src/model.py:321
    y_true[i] = x_full[b * (T + 1) + t + 1];    // x[:, 1:]    src/model.py:319
  }
}

}  // namespace

namespace wnp {

BlockPtrs block_ptrs(const wn_plan* p, int b, const float* params, const float* fragbase, int B, int T) {
  BlockPtrs k;
  memset(&k, 0, sizeof(k));
  const BlockInfo& bi = p->blocks[b];
  k.B = B; k.T = T; k.KS = p->KS; k.R = p->R; k.D = p->D; k.S = p->S; k.Cin = p->R; k.depth = p->LPB;
  k.act = p->c.activation; k.residual = p->c.use_residual;
  for (int i = 0; i < p->LPB; ++i) {
    const ConvInfo& c = bi.dil[i];
    k.dil[i] = c.dil;
    k.Wd[i] = params + p->tensors[c.kernel_t].off;
    k.bd[i] = params + p->tensors[c.bias_t].off;
    k.Fd[i] = fragbase + c.fragF; k.Fd_stride[i] = c.fragF_stride;
    k.Bd[i] = fragbase + c.fragB; k.Bd_stride[i] = c.fragB_stride;
  }
  k.br = params + p->tensors[bi.conv1.bias_t].off;
  k.Fr = fragbase + bi.conv1.fragF; k.Br_ = fragbase + bi.conv1.fragB;
  if (bi.has_skip) { k.bs = params + p->tensors[bi.conv_skip.bias_t].off; k.Bs = fragbase + bi.conv_skip.fragB; }
  k.Cc = 0; k.cond = nullptr; k.cb = nullptr;
  k.fused = p->fused_ok;
  if (p->fused16_ok && p->LPB == 1) { k.F16d = fragbase + bi.dil.back().frag16; k.F16r = fragbase + bi.conv1.frag16; }
  if (bi.f16gate >= 0) { k.F16g = fragbase + bi.f16gate; k.F16r = fragbase + bi.conv1.frag16; }
  if (bi.f16nat >= 0) k.F16n = fragbase + bi.f16nat;
  if (bi.g16u >= 0 && p->LPB == 1) k.G16u = fragbase + bi.g16u;
  if (bi.g16uf >= 0) k.G16uf = fragbase + bi.g16uf;
  if (p->LPB == 1 && bi.dil[0].frag16B >= 0) k.G16x = fragbase + bi.dil[0].frag16B;
  return k;
}

void deep16_ptrs(const wn_plan* p, int b, const float* fragbase, BlockPtrs& k) {
  if (!deep16(p)) return;
  const BlockInfo& bi = p->blocks[b];
  k.F16d = fragbase + bi.dil.back().frag16; k.F16r = fragbase + bi.conv1.frag16;
  for (int i = 0; i < p->LPB; ++i) {
    const ConvInfo& c = bi.dil[i];
    if (bi.d16F[i] >= 0) { k.F16i[i] = fragbase + bi.d16F[i]; k.JTi[i] = std::max(2, ceil32(c.cout)); }
    if (bi.d16B[i] >= 0) { k.G16i[i] = fragbase + bi.d16B[i]; k.JTb[i] = std::max(2, ceil32(c.cin)); }
  }
  if (bi.g16u >= 0) { k.G16u = fragbase + bi.g16u; k.JTu = std::max(2, ceil32(p->D)); }
}

int ring_capture(const float* src, int B, int T, int C, int nslots, float* ring, hipStream_t s) {
  const int64_t n = (int64_t)nslots * B * C;
  hipLaunchKernelGGL(wn_ring_capture_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s,
                     src, B, T, C, nslots, ring);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int forward_core(wn_plan* p, const float* params, const float* x, bool prep, const float* cond, int B,
                 int T, bool training, float* ws, const WsLayout& L, hipStream_t s, const GenRings* rings, LossFuse* lf) {
  int rc = ensure_device_tables(p);
  if (rc) return rc;
  const int64_t rows = (int64_t)B * T;
  float* fragbase = ws + L.frag;
  if (prep) {
    rc = wn_launch_prep_table(p->d_prep, (int)p->prep.size(), params, fragbase, s);
    if (rc) return rc;
  }
  // The weight-space preparation of the skip path (bias sum, V = W_s W_f0, its fp16 images: ~45 us of small launches) is
  // only needed by the folded contraction at the END of the block chain.  With the side stream allowed (knob 9 = 0) it is
  // forked off right where the stack begins (the stack's start event below) and runs beside the first blocks; the
  // contraction waits for it.  Otherwise it runs here, on the caller's stream.
  wn_exec& exs = wnp::ex(p);
  const bool fold = fold_ok(p);
  const bool fork_prep = prep && fold && p->c.use_skip && wn_debug_get(9) == 0;
  bool prep_forked = false;
  auto fold_prep = [&](hipStream_t sp) -> int {
    const bool fp_prof = prep && fold && exs.foldprep_used + 2 <= (int)exs.foldprep_ev.size();
    if (fp_prof) (void)hipEventRecord(exs.foldprep_ev[exs.foldprep_used], sp);
    // bias of the folded skip sum = sum over blocks of conv_skip (or conv1) biases
    if (prep && p->c.use_skip) {
      const ConvInfo& c0 = p->blocks[0].has_skip ? p->blocks[0].conv_skip : p->blocks[0].conv1;
      WnVecSumArgs v;
      v.base = params; v.off0 = p->tensors[c0.bias_t].off;
      v.stride = p->N > 1 ? (p->tensors[(p->blocks[1].has_skip ? p->blocks[1].conv_skip : p->blocks[1].conv1).bias_t].off - v.off0) : 0;
      v.count = p->N; v.len = p->Sh; v.out = ws + L.bias_sum;
      const int r = wn_launch_vecsum(v, sp);
      if (r) return r;
    }
    // (inference and the generation priming pass fold too: the queued sampler carries the folded contraction in its chain
    // kernel and must reproduce the sliding window bit for bit)
    if (fold && prep) {
      const BlockInfo& b0 = p->blocks[0];
      const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.kernel_t].off - p->tensors[b0.conv_skip.kernel_t].off : 0;
      int r = wn_launch_skip_fold(params, p->tensors[b0.conv_skip.kernel_t].off, wst, p->tensors[p->finals[0].kernel_t].off,
                                  p->tensors[p->finals[0].bias_t].off, ws + L.bias_sum, p->N, p->D, p->S, p->fold_F0, ws + L.vfold,
                                  ws + L.bfold, ws + L.wsall, sp);
      if (r) return r;
      r = wn_launch_prep_table(p->d_prep2, (int)p->prep2.size(), ws, fragbase, sp, 64);  // sources relative to the workspace
      if (r) return r;
    }
    if (fp_prof) { (void)hipEventRecord(exs.foldprep_ev[exs.foldprep_used + 1], sp); exs.foldprep_used += 2; }
    return WN_OK;
  };
  if (!fork_prep) {
    rc = fold_prep(s);
    if (rc) return rc;
  }
  // conditioning: mapping Dense stack + per-block time-invariant bias  (src/model.py:221-225,
  // src/layers.py:203-204: conv_cond(repeat(m)) == per-utterance bias)
  const float* m = cond;
  if (p->c.cond_inputs > 0) {
    if (!cond) { wn_set_error("Conditioning must be provided."); return WN_E_INVALID; }
    int mc = p->c.cond_inputs;
    for (size_t j = 0; j < p->mapping.size(); ++j) {
      const ConvInfo& c = p->mapping[j];
      if (cond_small(p))      // Dense: M[j] = act(m W + b), W = kernel (cin, cout)
        rc = wn_launch_sgemm_small_batched(m, mc, 1, 0, params + p->tensors[c.kernel_t].off, c.cout, 1, 0, ws + L.M[j], c.cout, 0,
                                           B, c.cout, mc, 1, params + p->tensors[c.bias_t].off, p->c.mapping_activation, s);
      else
      rc = Gemm(1, B, c.cout, ceil32(c.cout)).seg(m, mc, mc, 0, fragbase + c.fragF)
               .bias(params + p->tensors[c.bias_t].off).act(p->c.mapping_activation).run(ws + L.M[j], c.cout, s);
      if (rc) return rc;
      m = ws + L.M[j]; mc = c.cout;
    }
    if (p->frag_condF >= 0) {
      // all blocks in one contraction, then [B][N*2D] -> [N][B][2D] with the biases added
      const int D2 = 2 * p->D;
      const ConvInfo& c0 = p->blocks[0].conv_cond;
      const int64_t bst = p->N > 1 ? p->tensors[p->blocks[1].conv_cond.bias_t].off - p->tensors[c0.bias_t].off : 0;
      if (cond_small(p)) {    // block z: cbt[:, z * 2D ..] = m W_c(z), W_c = kernel (1, Cc, 2D)
        const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_cond.kernel_t].off - p->tensors[c0.kernel_t].off : 0;
        rc = wn_launch_sgemm_small_batched(m, p->Cc, 1, 0, params + p->tensors[c0.kernel_t].off, D2, 1, wst, ws + L.cbt, p->N * D2, D2,
                                           B, D2, p->Cc, p->N, nullptr, 0, s);
      } else
      rc = Gemm(1, B, p->N * D2, ceil32(p->N * D2)).seg(m, p->Cc, p->Cc, 0, fragbase + p->frag_condF).run(ws + L.cbt, p->N * D2, s);
      if (rc) return rc;
      rc = wn_launch_cond_scatter(ws + L.cbt, params, p->tensors[c0.bias_t].off, bst, B, p->N, D2, ws + L.cb, s);
      if (rc) return rc;
    } else {
    for (int b = 0; b < p->N; ++b) {
      const ConvInfo& c = p->blocks[b].conv_cond;
      rc = Gemm(1, B, 2 * p->D, ceil32(2 * p->D)).seg(m, p->Cc, p->Cc, 0, fragbase + c.fragF)
               .bias(params + p->tensors[c.bias_t].off).run(ws + L.cb + (int64_t)b * B * 2 * p->D, 2 * p->D, s);
      if (rc) return rc;
    }
    }
  }
  // Forward range guard.  The split-precision kernels cast fp32 activations to fp16 hi|lo unscaled: beyond 65504 the
  // hi part is inf.  Every kernel that produces an input of such a kernel -- the residual stream H[b], the skip sum,
  // the head activations (z is bounded by 1) -- publishes its running max-abs here; the callers turn it into a flag
  // (WN_RANGE_LIMIT) and redo the pass with the exact-fp32 kernels when it tripped.
  float* const fam = ws + L.fwd_absmax;
  WN_HIP_CHECK(hipMemsetAsync(fam, 0, sizeof(float), s));
  // input causal conv, src/model.py:84-88,228 : KS taps with C_in = 1
  {
    if (p->R % 4 == 0 && wn_debug_get(1) != 1) {
      // elementwise kernel, same fma chain as the matrix product below computes for a K = 1 operand
      rc = wn_launch_inconv_fwd(x, params + p->tensors[p->causal.kernel_t].off, params + p->tensors[p->causal.bias_t].off, B, T,
                                p->R, p->KS, ws + L.H[0], fam, s);
    } else {
      Gemm g(B, T, p->R, ceil32(p->R));
      for (int t = 0; t < p->KS; ++t)
        g.seg(x, 1, 1, (p->KS - 1 - t), fragbase + p->causal.fragF + t * p->causal.fragF_stride);
      rc = g.bias(params + p->tensors[p->causal.bias_t].off).run(ws + L.H[0], p->R, s);
    }
    if (rc) return rc;
    if (rings) {
      rc = ring_capture(x, B, T, 1, p->KS, rings->xin, s);
      if (!rc) rc = ring_capture(ws + L.H[0], B, T, p->R, rings->nslots[0], rings->h[0], s);
      if (rc) return rc;
    }
  }
  if (p->Dp != p->D) {
    rc = wn_launch_fill(ws + L.Z, 0.f, rows * p->N * p->Dp, s);
    if (rc) return rc;
  }
  // residual blocks, src/model.py:230-234
  // profiling: is the chain N back-to-back launches of the fused block kernel?
  const bool prof_chain = wnp::ex(p).prof_on && !rings && p->LPB == 1 && p->c.cond_inputs == 0 && p->R == p->D &&
                          !(training && wnp::ex(p).drop_rate > 0.f) && block_ptrs(p, 0, params, fragbase, B, T).fused;
  const bool stack_prof = !rings && wnp::ex(p).stack_used + 2 <= (int)wnp::ex(p).stack_ev.size();
  if (stack_prof) (void)hipEventRecord(wnp::ex(p).stack_ev[wnp::ex(p).stack_used], s);
  if (fork_prep) {
    // the stack's start event (above) is the fork point: whatever the preparation costs the chain shows inside the pair
    if (!exs.side) {
      WN_HIP_CHECK(hipStreamCreateWithFlags(&exs.side, hipStreamNonBlocking));
      WN_HIP_CHECK(hipEventCreateWithFlags(&exs.ev_fork, hipEventDisableTiming));
      WN_HIP_CHECK(hipEventCreateWithFlags(&exs.ev_join, hipEventDisableTiming));
    }
    if (!exs.ev_ffork) {
      WN_HIP_CHECK(hipEventCreateWithFlags(&exs.ev_ffork, hipEventDisableTiming));
      WN_HIP_CHECK(hipEventCreateWithFlags(&exs.ev_fjoin, hipEventDisableTiming));
    }
    WN_HIP_CHECK(hipEventRecord(exs.ev_ffork, s));     // behind whatever wrote the parameters
    WN_HIP_CHECK(hipStreamWaitEvent(exs.side, exs.ev_ffork, 0));
    rc = fold_prep(exs.side);
    if (rc) return rc;
    WN_HIP_CHECK(hipEventRecord(exs.ev_fjoin, exs.side));
    prep_forked = true;
  }
  for (int b = 0; b < p->N; ++b) {
    BlockPtrs k = block_ptrs(p, b, params, fragbase, B, T);
    if (training && !rings) deep16_ptrs(p, b, fragbase, k);
    if (p->c.cond_inputs > 0) k.cb = ws + L.cb + (int64_t)b * B * 2 * p->D;
    BlockBufs f;
    memset(&f, 0, sizeof(f));
    const int hi = training ? b : (b & 1), ho = training ? b + 1 : ((b + 1) & 1);
    f.x = ws + L.H[hi];
    if (training && wnp::ex(p).drop_rate > 0.f) {
      // x = dropout(x) feeds the dilated stack; the residual keeps the original (src/layers.py:192-196)
      rc = wn_launch_dropout(ws + L.H[hi], nullptr, ws + L.XD[b], rows * p->R, wnp::ex(p).drop_rate,
                             wn_dropout_key(wnp::ex(p).drop_seed, b, wnp::ex(p).drop_step), nullptr, s);
      if (rc) return rc;
      f.x = ws + L.XD[b];
      f.res = ws + L.H[hi];
    }
    for (int i = 0; i + 1 < p->LPB; ++i) f.P[i] = ws + L.P[b][i];
    f.U = ws + L.U;
    f.AG = training ? ws + L.AG[b] : nullptr;
    f.Z = ws + L.Z + (int64_t)b * rows * p->Dp; f.ldz = p->Dp;      // block-major [N][rows][Dp]
    f.O = nullptr;
    f.x_out = ws + L.H[ho];
    f.fwd_absmax = fam;
    const bool prof = wnp::ex(p).prof_on && wnp::ex(p).prof_used + 2 <= (int)wnp::ex(p).prof_ev.size();
    const bool ev0 = prof && (!prof_chain || b == 0), ev1 = prof && (!prof_chain || b == p->N - 1);
    if (ev0) (void)hipEventRecord(wnp::ex(p).prof_ev[wnp::ex(p).prof_used], s);
    rc = block_forward(k, f, s);
    if (rc) return rc;
    if (ev1) {
      (void)hipEventRecord(wnp::ex(p).prof_ev[wnp::ex(p).prof_used + 1], s);
      wnp::ex(p).prof_cnt[wnp::ex(p).prof_used / 2] = prof_chain ? p->N : 1;
      wnp::ex(p).prof_used += 2;
    }
    if (rings && b + 1 < p->N) {
      rc = ring_capture(f.x_out, B, T, p->R, rings->nslots[b + 1], rings->h[b + 1], s);
      if (rc) return rc;
    }
    if (rings)
      for (int i = 0; i + 1 < p->LPB; ++i) {
        rc = ring_capture(f.P[i], B, T, p->D, rings->nslots_p[b][i], rings->hp[b][i], s);
        if (rc) return rc;
      }
  }
  if (prep_forked) WN_HIP_CHECK(hipStreamWaitEvent(s, exs.ev_fjoin, 0));   // the fold's images and biases are ready
  // skip sum folded into one contraction over all blocks' gated activations (src/model.py:235-236
  // with src/layers.py:216-219), or the last block output when use_skip is False
  const float* hin;
  size_t first_final = 0;
  if (fold) {
    // a = act(sum_b V(b)^T z_b + b'): the skip sum and the head's first conv in ONE contraction with F0 output columns
    const ConvInfo& c0 = p->finals[0];
    // streamed kernel, second form (wn_gemm16s.hip; other shapes: wn_gemm_rows16_kernel, the same products in the same order)
    // (the streamed form indexes rows with 32-bit byte offsets: beyond 4 GiB per plane the rows GEMM takes over)
    if (p->Dp == p->D && wn_gemm_planes16s_supported(c0.cout, p->D, p->N, p->Dp, c0.cout) &&
        (int64_t)rows * p->Dp * 4 < ((int64_t)1 << 32) && (int64_t)rows * c0.cout * 4 < ((int64_t)1 << 32)) {
      WnGemmPlanesArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.z = ws + L.Z; ga.plane_stride = rows * p->Dp; ga.ld = p->Dp; ga.plane_k = p->D; ga.nplanes = p->N;
      ga.w16 = fragbase + p->frag16_foldF; ga.bias = ws + L.bfold; ga.act = p->c.activation;
      ga.y = ws + L.HA[0]; ga.ldy = c0.cout; ga.N = c0.cout; ga.B = B; ga.T = T; ga.absmax_out = fam;
      rc = wn_launch_gemm_planes16s(ga, s);
    } else
    rc = Gemm(B, T, c0.cout, ceil32(c0.cout)).seg_planes(ws + L.Z, p->Dp, rows * p->Dp, p->N * p->Dp, nullptr)
             .w16(fragbase + p->frag16_foldF).bias(ws + L.bfold).act(p->c.activation).absmax_fwd(fam)
             .run(ws + L.HA[0], c0.cout, s);
    if (rc) return rc;
    hin = ws + L.HA[0];
    first_final = 1;
  } else if (p->c.use_skip) {
    rc = Gemm(B, T, p->Sh, ceil32(p->Sh)).seg_planes(ws + L.Z, p->Dp, rows * p->Dp, p->N * p->Dp, fragbase + p->frag_skipF)
             .w16(p->frag16_skipF >= 0 ? fragbase + p->frag16_skipF : nullptr)
             .bias(ws + L.bias_sum).absmax_fwd(fam).run(ws + L.skipsum, p->Sh, s);
    if (rc) return rc;
    hin = ws + L.skipsum;
  } else {
    hin = ws + L.H[training ? p->N : (p->N & 1)];
  }
  if (stack_prof) { (void)hipEventRecord(wnp::ex(p).stack_ev[wnp::ex(p).stack_used + 1], s); wnp::ex(p).stack_used += 2; }
  // head, src/model.py:105-119,237-238: conv -> activation, last conv linear (softmax applied later)
  int hc = fold ? p->finals[0].cout : p->Hin;
  for (size_t i = first_final; i < p->finals.size(); ++i) {
    const ConvInfo& c = p->finals[i];
    const bool last = (i + 1 == p->finals.size());
    float* dst = last ? ws + L.logits : ws + L.HA[i];
    // training pass of a 256-class categorical head: the loss is this conv's epilogue (no logits tensor)
    if (last && lf && loss_fusable(p, rows) && wn_gemm_planes16s_supported(c.cout, hc, 1, hc, c.cout) &&
        (int64_t)rows * hc * 4 < ((int64_t)1 << 32)) {
      WnGemmPlanesArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.z = hin; ga.plane_stride = 0; ga.ld = hc; ga.plane_k = hc; ga.nplanes = 1;
      ga.w16 = fragbase + c.frag16; ga.bias = params + p->tensors[c.bias_t].off; ga.act = WN_ACT_LINEAR;
      ga.y = lf->g_logits; ga.ldy = c.cout; ga.N = c.cout; ga.B = B; ga.T = T; ga.absmax_out = lf->absmax_out;
      ga.cat_loss = 1; ga.target = lf->target; ga.gscale = lf->gscale; ga.loss_rows = lf->loss_rows;
      ga.sample_out = lf->sample_out; ga.inv_lv = lf->inv_lv; ga.seed = lf->seed; ga.offset = lf->offset;
      rc = wn_launch_gemm_planes16s(ga, s);
      if (rc) return rc;
      lf->done = true;
      break;
    }
    // 128 / 256 output columns: the streamed kernel's second form (wn_gemm16s.hip; the operand is one "plane"); same
    // products in the same order as the rows GEMM below
    if (c.frag16 >= 0 && wn_debug_get(1) != 1 && wn_gemm_planes16s_supported(c.cout, hc, 1, hc, c.cout) &&
        (int64_t)rows * hc * 4 < ((int64_t)1 << 32)) {
      WnGemmPlanesArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.z = hin; ga.plane_stride = 0; ga.ld = hc; ga.plane_k = hc; ga.nplanes = 1;
      ga.w16 = fragbase + c.frag16; ga.bias = params + p->tensors[c.bias_t].off; ga.act = last ? WN_ACT_LINEAR : p->c.activation;
      ga.y = dst; ga.ldy = c.cout; ga.N = c.cout; ga.B = B; ga.T = T; ga.absmax_out = last ? nullptr : fam;
      rc = wn_launch_gemm_planes16s(ga, s);
      if (rc) return rc;
      hin = dst; hc = c.cout;
      continue;
    }
    rc = Gemm(B, T, c.cout, ceil32(c.cout)).seg(hin, hc, hc, 0, fragbase + c.fragF)
             .w16(c.frag16 >= 0 ? fragbase + c.frag16 : nullptr)
             .bias(params + p->tensors[c.bias_t].off).act(last ? WN_ACT_LINEAR : p->c.activation)
             .absmax_fwd(last ? nullptr : fam).run(dst, c.cout, s);
    if (rc) return rc;
    hin = dst; hc = c.cout;
  }
  return WN_OK;
}

int shift_split(const float* x_full, int B, int T, float* inputs, float* y_true, hipStream_t s) {
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(wn_shift_split_kernel, dim3((unsigned)std::min<int64_t>((rows + 255) / 256, 4096)), dim3(256), 0, s,
                     x_full, B, T, inputs, y_true);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

// the split-precision head of a 256-class categorical model whose last conv the streamed 256-column kernel takes
bool loss_fusable(const wn_plan* p, int64_t rows) {
  if (p->c.head != WN_HEAD_CATEGORICAL || p->Cout != 256 || wn_debug_get(1) == 1 || p->finals.empty()) return false;
  const ConvInfo& c = p->finals.back();
  return c.frag16 >= 0 && c.cout == 256 && rows > 0;
}

int loss_stage(wn_plan* p, int B, int T, int global_batch, bool want_grad, float* ws, const WsLayout& L,
               float* loss_out, float* absmax_out, hipStream_t s, bool fused_done) {
  const int64_t rows = (int64_t)B * T;
  const float gscale = 1.0f / (float)global_batch;     // compute_average_loss, src/model.py:328-329
  int rc;
  // (the head's last conv already left the row losses and d loss / d logits: see LossFuse)
  if (fused_done) return wn_launch_sum(ws + L.loss_rows, rows, gscale, loss_out, ws + L.sum_scratch, s);
  // deferred weight gradients read d loss / d logits from GF.back(): written there directly (no 131 MB copy)
  float* g_logits = want_grad ? ws + L.GF.back() : nullptr;
  if (p->c.head == WN_HEAD_CATEGORICAL) {
    rc = wn_launch_quantize(ws + L.yt, reinterpret_cast<int32_t*>(ws + L.target), rows, p->c.bits, s);
    if (rc) return rc;
    // an armed step sample (wn_plan_arm_step_sample) rides in the loss kernel when the row fits its registers
    float* so = nullptr;
    if (want_grad && wnp::ex(p).step_sample && !wnp::ex(p).step_sample_det && p->Cout <= 256) { so = wnp::ex(p).step_sample; wnp::ex(p).step_sample = nullptr; }
    rc = wn_launch_cat_loss(ws + L.logits, reinterpret_cast<const int32_t*>(ws + L.target), rows, p->Cout,
                            gscale, ws + L.loss_rows, g_logits, absmax_out, s, so, p->c.bits, wnp::ex(p).step_sample_seed,
                            wnp::ex(p).step_sample_off);
  } else {
    rc = wn_launch_mix_loss(ws + L.logits, ws + L.yt, rows, p->c.num_mixtures, p->c.bits,
                            p->c.head == WN_HEAD_LOGISTIC ? 1 : 2, gscale, ws + L.loss_rows, g_logits, absmax_out, s);
  }
  if (rc) return rc;
  return wn_launch_sum(ws + L.loss_rows, rows, gscale, loss_out, ws + L.sum_scratch, s);
}

}  // namespace wnp

using namespace wnp;

extern "C" int wn_forward(wn_plan* p, const float* params, const float* x, const float* cond, int32_t B,
                          int32_t T, float* out, float* logits_out, float* workspace, int64_t ws_floats,
                          void* stream) {
  if (!p || !params || !x || !workspace || B < 1 || T < 1) { wn_set_error("forward: bad arguments"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const WsLayout L = make_layout(p, B, T, false);
  if (ws_floats < L.total) { wn_set_error("forward: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)L.total); return WN_E_INVALID; }
  int rc = forward_core(p, params, x, true, cond, B, T, false, workspace, L, s);
  if (rc) return rc;
  const int64_t rows = (int64_t)B * T;
  if (logits_out) WN_HIP_CHECK(hipMemcpyAsync(logits_out, workspace + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (out) {
    if (p->c.head == WN_HEAD_CATEGORICAL) return wn_launch_softmax(workspace + L.logits, out, rows, p->Cout, s);
    WN_HIP_CHECK(hipMemcpyAsync(out, workspace + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return WN_OK;
}

// WaveNet.call(inputs, training=True), src/model.py:213-239 with src/layers.py:195-196: the forward pass with the
// Dropout layers active (the mask of the step set by wn_plan_set_dropout).  Needs the TRAINING workspace size.
extern "C" int wn_forward_training(wn_plan* p, const float* params, const float* x, const float* cond, int32_t B,
                                   int32_t T, float* out, float* logits_out, float* workspace, int64_t ws_floats,
                                   void* stream) {
  if (!p || !params || !x || !workspace || B < 1 || T < 1) { wn_set_error("forward_training: bad arguments"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const WsLayout L = make_layout(p, B, T, true);
  if (ws_floats < L.total) { wn_set_error("forward_training: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)L.total); return WN_E_INVALID; }
  int rc = forward_core(p, params, x, true, cond, B, T, true, workspace, L, s);
  if (rc) return rc;
  const int64_t rows = (int64_t)B * T;
  if (logits_out) WN_HIP_CHECK(hipMemcpyAsync(logits_out, workspace + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (out) {
    if (p->c.head == WN_HEAD_CATEGORICAL) return wn_launch_softmax(workspace + L.logits, out, rows, p->Cout, s);
    WN_HIP_CHECK(hipMemcpyAsync(out, workspace + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return WN_OK;
}

extern "C" int wn_eval_loss(wn_plan* p, const float* params, const float* x_full, const float* cond,
                            int32_t B, int32_t T, int32_t global_batch, float* loss_out, float* pred_out,
                            float* workspace, int64_t ws_floats, void* stream) {
  if (!p || !params || !x_full || !workspace || !loss_out || B < 1 || T < 1) { wn_set_error("eval_loss: bad arguments"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const WsLayout L = make_layout(p, B, T, false);
  if (ws_floats < L.total) { wn_set_error("eval_loss: workspace too small"); return WN_E_INVALID; }
  const int64_t rows = (int64_t)B * T;
  // inputs live in the (otherwise unused here) probs region
  float* inputs = workspace + L.probs;
  { const int rcs = shift_split(x_full, B, T, inputs, workspace + L.yt, s); if (rcs) return rcs; }
  int rc = forward_core(p, params, inputs, true, cond, B, T, false, workspace, L, s);
  if (rc) return rc;
  rc = loss_stage(p, B, T, global_batch > 0 ? global_batch : B, false, workspace, L, loss_out, nullptr, s);
  if (rc) return rc;
  rc = wn_launch_guard_flag(workspace + L.fwd_absmax, WN_RANGE_LIMIT, wn_debug_get(1) != 1, loss_out + 2, s);
  if (rc) return rc;
  if (pred_out) {
    if (p->c.head == WN_HEAD_CATEGORICAL) return wn_launch_softmax(workspace + L.logits, pred_out, rows, p->Cout, s);
    WN_HIP_CHECK(hipMemcpyAsync(pred_out, workspace + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return WN_OK;
}
