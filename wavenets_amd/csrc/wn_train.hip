// The gradient half of WaveNet.train_step (src/model.py:319-336): backward-data chain, deferred weight gradients, and the
// optimizer step (Adam with per-tensor clipnorm, train.py:225-226).
#include "wn_plan_internal.h"

namespace {
using namespace wnp;

// ---- batched weight-gradient job table for one (B, T) layout ----
void add_jobs(std::vector<WnWgJob>& jobs, int64_t x_off, int ldx, int K, int shift, int64_t g_off, int ldg, int N,
              int64_t out_off, int64_t bias_off, int64_t gmax_off) {
  const int tk = wn_wgrad_tile_k(), tn = wn_wgrad_tile_n();
  for (int k0 = 0; k0 < K; k0 += tk)
    for (int n0 = 0; n0 < N; n0 += tn) {
      WnWgJob j;
      memset(&j, 0, sizeof(j));
      j.x_off = x_off; j.g_off = g_off; j.out_off = out_off; j.bias_off = (k0 == 0) ? bias_off : -1;
      j.gmax_off = gmax_off;
      j.ldx = ldx; j.ldg = ldg; j.K = K; j.N = N; j.shift = shift; j.k0 = k0; j.n0 = n0;
      jobs.push_back(j);
    }
}

// the dedicated skip weight-gradient kernel applies to the split-precision path with uniform blocks
bool skip_kernel_ok(const wn_plan* p) {
  return p->c.use_skip && p->S > 0 && p->Dp == p->D && p->N >= 1 && wn_debug_get(1) != 1 &&
         wn_wgrad_skip_supported(p->D, p->S, p->N * p->D);
}

int ensure_jobs(wn_plan* p, const WsLayout& L, int B, int T) {
  const bool skipk = skip_kernel_ok(p);
  // 32 / 64-channel blocks: one workgroup per (block, utterance, time range) (wn_wgrad_layer.hip; stacks deeper than 1 in
  // split-precision training, deep16: the last conv + the 1x1 as for depth 1, every inner conv through the kernel's INNER
  // form); other shapes stay on the generic job table
  const bool layerk = (p->LPB == 1 || deep16(p)) && wn_wgrad_layer_supported(p->R, p->D, p->KS) && p->Dp == p->R && wn_debug_get(1) != 1;
  // 128-channel blocks: two transposed-LDS-read jobs per block (wn_wgrad_tr.hip): both taps of the gated conv against ONE
  // read of du; dW_r
  const bool pairk = !layerk && p->LPB == 1 && p->KS == 2 && p->R == p->D && p->Dp == p->D && wn_wgrad_pair_kind(p->R, 2 * p->D) == 1 &&
                     wn_wgrad_pair_kind(p->D, p->R) == 2 && wn_debug_get(1) != 1;
  // ... whose dW_r jobs also compute the folded skip path's M = Z^T dL/da (one read of z)
  const bool mfused = pairk && p->D == 128 && fold_ok(p) && p->fold_F0 == 128 && p->Dp == p->D && p->S > 0;
  // 64- / 32-channel blocks: M as transposed-read jobs over the z of four / eight blocks at a time against one read of
  // dL/da (wn_wgrad_tr kinds 7 / 8); other widths: wn_wgrad_skip_kernel
  const int mtr = (layerk && fold_ok(p) && p->fold_F0 == 128 && p->Dp == p->D && p->S > 0)
                      ? (p->D == 64 ? 7 : (p->D == 32 ? 8 : 0)) : 0;
  const bool fold = fold_ok(p);
  const bool headpairs = head_pairs_ok(p) && L.hsplits > 0;
  const bool inconvk = L.isplits > 0;
  const bool d16 = deep16(p);
  if (wnp::ex(p).d_jobs && wnp::ex(p).jobs_B == B && wnp::ex(p).jobs_T == T && wnp::ex(p).jobs_splits == L.bsplits &&
      wnp::ex(p).jobs_drop == (wnp::ex(p).drop_rate > 0.f) && wnp::ex(p).jobs_skipk == skipk && wnp::ex(p).jobs_layerk == layerk &&
      wnp::ex(p).jobs_pairk == pairk && wnp::ex(p).jobs_mfused == mfused && wnp::ex(p).jobs_mtr == mtr && wnp::ex(p).jobs_deep16 == d16 && wnp::ex(p).jobs_headpairs == headpairs && wnp::ex(p).jobs_inconvk == inconvk && wnp::ex(p).jobs_fold == fold) return WN_OK;
  std::vector<WnWgLayer> wgl, wgli;
  std::vector<WnWgPair> pairs[3];
  std::vector<WnWgPair> hpairs[6];
  std::vector<WnWgJob> jobs;
  std::vector<WnTensorDesc> cov;
  auto cover = [&](int t) { WnTensorDesc d; d.off = p->tensors[t].off; d.len = p->tensors[t].len; cov.push_back(d); };
  // running max-abs slots (same numbering as in wn_train_fwd_bwd): GF[i] | g_skipsum | GU[b] | GH[b]
  const int nfin = (int)p->finals.size();
  const int64_t am_skip = L.absmax + nfin;
  auto am_GU = [&](int b) { return L.absmax + nfin + 1 + b; };
  auto am_GH = [&](int b) { return L.absmax + nfin + 1 + p->N + b; };
  auto am_GP = [&](int b, int i) { return d16 ? L.absmax + nfin + 1 + p->N + (p->N + 1) + (int64_t)b * (p->LPB - 1) + i : (int64_t)-1; };
  // input causal conv: x = inputs (B,T,1), g = d loss / d H[0]
  if (!inconvk)
    for (int t = 0; t < p->KS; ++t)
      add_jobs(jobs, L.probs, 1, 1, p->KS - 1 - t, L.GH[0], p->R, p->R,
               p->tensors[p->causal.kernel_t].off + (int64_t)t * p->R,
               t == p->KS - 1 ? p->tensors[p->causal.bias_t].off : -1, am_GH(0));
  cover(p->causal.kernel_t); cover(p->causal.bias_t);
  for (int b = 0; b < p->N; ++b) {
    const BlockInfo& bi = p->blocks[b];
    const ConvInfo& c = bi.dil.back();
    const int64_t zoff = L.Z + (int64_t)b * B * T * p->Dp;      // block-major Z
    if (layerk) {
      const int64_t xin0 = wnp::ex(p).drop_rate > 0.f ? L.XD[b] : L.H[b];
      for (int i = 0; i + 1 < p->LPB; ++i) {                     // inner convs of a deeper stack
        const ConvInfo& ci = bi.dil[i];
        WnWgLayer w;
        memset(&w, 0, sizeof(w));
        w.x_off = i == 0 ? xin0 : L.P[b][i - 1];
        w.du_off = L.GP[b][i];
        w.dwd_off = p->tensors[ci.kernel_t].off; w.dbd_off = p->tensors[ci.bias_t].off;
        w.dwr_off = w.dbr_off = -1; w.z_off = w.go_off = 0;
        w.gmax_u_off = am_GP(b, i); w.gmax_h_off = -1;
        w.dilation = ci.dil; w.ldz = p->Dp;
        wgli.push_back(w);
        cover(ci.kernel_t); cover(ci.bias_t);
      }
      WnWgLayer w;
      w.x_off = p->LPB > 1 ? L.P[b][p->LPB - 2] : xin0;
      w.du_off = L.GU[b]; w.z_off = zoff; w.ldz = p->Dp;
      w.go_off = p->S == 0 ? L.GO[b] : L.GH[b + 1];
      w.dwd_off = p->tensors[c.kernel_t].off; w.dbd_off = p->tensors[c.bias_t].off;
      w.dwr_off = p->tensors[bi.conv1.kernel_t].off; w.dbr_off = p->tensors[bi.conv1.bias_t].off;
      w.gmax_u_off = am_GU(b); w.gmax_h_off = p->S == 0 ? am_skip : am_GH(b + 1);
      w.dilation = c.dil;
      wgl.push_back(w);
    } else if (pairk) {
      const int64_t xoff = wnp::ex(p).drop_rate > 0.f ? L.XD[b] : L.H[b];
      {
        // both taps in one job: x[t - d] | x[t] against ONE read of du
        WnWgPair w;
        memset(&w, 0, sizeof(w));
        w.x_off = xoff; w.g_off = L.GU[b]; w.shift = c.dil;
        w.w_off = p->tensors[c.kernel_t].off;
        w.b_off = p->tensors[c.bias_t].off;
        w.gmax_off = am_GU(b);
        pairs[1].push_back(w);
      }
      WnWgPair w;
      memset(&w, 0, sizeof(w));
      w.x_off = zoff; w.g_off = p->S == 0 ? L.GO[b] : L.GH[b + 1]; w.shift = 0;
      w.w_off = p->tensors[bi.conv1.kernel_t].off; w.b_off = p->tensors[bi.conv1.bias_t].off;
      w.gmax_off = p->S == 0 ? am_skip : am_GH(b + 1);
      w.g2_off = w.w2_off = w.b2_off = w.gmax2_off = -1;
      if (mfused) {
        // the folded skip path's M(b) = z_b^T dL/da rides in the same job (one read of z_b): second slab = mslab
        w.g2_off = L.GF[0]; w.w2_off = (int64_t)b * p->D * p->fold_F0;
        w.b2_off = b == 0 ? (int64_t)p->N * p->D * p->fold_F0 : -1;       // colsum(dL/da) once
        w.gmax2_off = L.absmax + 0;                                        // am_GF(0)
      }
      pairs[2].push_back(w);
    } else {
    // the dilated stack: conv i reads H[b] (or its dropped copy) / the activated output of conv i - 1; its output gradient
    // is GP[b][i], or GU[b] for the last, gated conv (2D wide).  (Inner gradients have no max-abs slot: stacks deeper
    // than 1 run this table in exact fp32, see the launch.)
    for (int i = 0; i < p->LPB; ++i) {
      const ConvInfo& ci = bi.dil[i];
      const bool lastc = i == p->LPB - 1;
      const int64_t xo = i == 0 ? (wnp::ex(p).drop_rate > 0.f ? L.XD[b] : L.H[b]) : L.P[b][i - 1];
      const int kc = i == 0 ? p->R : p->D, nc = lastc ? 2 * p->D : p->D;
      for (int t = 0; t < p->KS; ++t)
        add_jobs(jobs, xo, kc, kc, (p->KS - 1 - t) * ci.dil, lastc ? L.GU[b] : L.GP[b][i], nc, nc,
                 p->tensors[ci.kernel_t].off + (int64_t)t * kc * nc,
                 t == p->KS - 1 ? p->tensors[ci.bias_t].off : -1, lastc ? am_GU(b) : am_GP(b, i));
      if (!lastc) { cover(ci.kernel_t); cover(ci.bias_t); }
    }
    // S == 0: g_o = g_xout + g_skip (or a copy of g_skip): bounded by twice the larger max-abs -> no slot
    add_jobs(jobs, zoff, p->Dp, p->D, 0, p->S == 0 ? L.GO[b] : L.GH[b + 1], p->R, p->R,
             p->tensors[bi.conv1.kernel_t].off, p->tensors[bi.conv1.bias_t].off, p->S == 0 ? am_skip : am_GH(b + 1));
    }
    cover(bi.dil.back().kernel_t); cover(bi.dil.back().bias_t);
    cover(bi.conv1.kernel_t); cover(bi.conv1.bias_t);
    if (bi.has_skip && p->c.use_skip && !fold) {     // (folded: dW_s, db_s come out of M, see the weight-gradient phase)
      if (!skipk)
        add_jobs(jobs, zoff, p->Dp, p->D, 0, L.g_skipsum, p->S, p->S,
                 p->tensors[bi.conv_skip.kernel_t].off, p->tensors[bi.conv_skip.bias_t].off, am_skip);
      cover(bi.conv_skip.kernel_t); cover(bi.conv_skip.bias_t);
    }
  }
  if (mtr != 0) {
    const int per = 256 / p->D;                               // blocks per job
    for (int b0 = 0; b0 < p->N; b0 += per) {
      WnWgPair w;
      memset(&w, 0, sizeof(w));
      w.x_off = L.Z + (int64_t)b0 * B * T * p->Dp;            // block-major Z: segment stride = one block's plane
      w.g2_off = (int64_t)B * T * p->Dp;
      w.pad_ = std::min(per, p->N - b0);
      w.g_off = L.GF[0]; w.shift = 0;
      w.w_off = (int64_t)b0 * p->D * p->fold_F0;
      w.b_off = b0 == 0 ? (int64_t)p->N * p->D * p->fold_F0 : -1;       // colsum(dL/da) once
      w.gmax_off = L.absmax + 0;                              // am_GF(0)
      w.w2_off = w.b2_off = w.gmax2_off = -1;
      pairs[0].push_back(w);
    }
  }
  wnp::ex(p).head_first = (int)jobs.size();
  wnp::ex(p).cov_head_first = (int)cov.size();
  for (size_t i = fold ? 1 : 0; i < p->finals.size(); ++i) {      // (folded: the first conv's gradients come from M too)
    const ConvInfo& c = p->finals[i];
    const int64_t xin = (i == 0) ? (p->c.use_skip ? L.skipsum : L.H[p->N]) : L.HA[i - 1];
    if (headpairs && wn_wgrad_pair_kind(c.cin, c.cout) != 0) {
      WnWgPair w;
      memset(&w, 0, sizeof(w));
      w.x_off = xin; w.g_off = L.GF[i]; w.shift = 0;
      w.w_off = p->tensors[c.kernel_t].off; w.b_off = p->tensors[c.bias_t].off;
      w.gmax_off = L.absmax + (int64_t)i;
      const int kind = wn_wgrad_pair_kind(c.cin, c.cout);
      hpairs[kind].push_back(w);
      if (kind == 5) {                     // second 128-column half
        w.g_off += 128; w.w_off += 128; w.b_off += 128;
        hpairs[kind].push_back(w);
      }
    } else {
      add_jobs(jobs, xin, c.cin, c.cin, 0, L.GF[i], c.cout, c.cout, p->tensors[c.kernel_t].off,
               p->tensors[c.bias_t].off, L.absmax + (int64_t)i);
    }
    cover(c.kernel_t); cover(c.bias_t);
  }
  if (wnp::ex(p).d_jobs) { (void)hipFree(wnp::ex(p).d_jobs); wnp::ex(p).d_jobs = nullptr; }
  if (wnp::ex(p).d_cov) { (void)hipFree(wnp::ex(p).d_cov); wnp::ex(p).d_cov = nullptr; }
  WN_HIP_CHECK(hipMalloc((void**)&wnp::ex(p).d_jobs, std::max<size_t>(jobs.size(), 1) * sizeof(WnWgJob)));
  if (!jobs.empty()) WN_HIP_CHECK(hipMemcpy(wnp::ex(p).d_jobs, jobs.data(), jobs.size() * sizeof(WnWgJob), hipMemcpyHostToDevice));
  WN_HIP_CHECK(hipMalloc((void**)&wnp::ex(p).d_cov, cov.size() * sizeof(WnTensorDesc)));
  WN_HIP_CHECK(hipMemcpy(wnp::ex(p).d_cov, cov.data(), cov.size() * sizeof(WnTensorDesc), hipMemcpyHostToDevice));
  wnp::ex(p).h_cov = cov;
  if (wnp::ex(p).d_wgl) { (void)hipFree(wnp::ex(p).d_wgl); wnp::ex(p).d_wgl = nullptr; }
  if (!wgl.empty()) {
    WN_HIP_CHECK(hipMalloc((void**)&wnp::ex(p).d_wgl, wgl.size() * sizeof(WnWgLayer)));
    WN_HIP_CHECK(hipMemcpy(wnp::ex(p).d_wgl, wgl.data(), wgl.size() * sizeof(WnWgLayer), hipMemcpyHostToDevice));
  }
  if (wnp::ex(p).d_wgli) { (void)hipFree(wnp::ex(p).d_wgli); wnp::ex(p).d_wgli = nullptr; }
  wnp::ex(p).n_wgli = (int)wgli.size();
  if (!wgli.empty()) {
    WN_HIP_CHECK(hipMalloc((void**)&wnp::ex(p).d_wgli, wgli.size() * sizeof(WnWgLayer)));
    WN_HIP_CHECK(hipMemcpy(wnp::ex(p).d_wgli, wgli.data(), wgli.size() * sizeof(WnWgLayer), hipMemcpyHostToDevice));
  }
  if (wnp::ex(p).d_pairs) { (void)hipFree(wnp::ex(p).d_pairs); wnp::ex(p).d_pairs = nullptr; }
  {
    std::vector<WnWgPair> all;
    for (int kd = 0; kd <= 2; ++kd) {
      wnp::ex(p).pair_first[kd] = (int)all.size();
      wnp::ex(p).pair_count[kd] = (int)pairs[kd].size();
      all.insert(all.end(), pairs[kd].begin(), pairs[kd].end());
    }
    for (int kd = 1; kd <= 5; ++kd) {
      wnp::ex(p).hpair_first[kd] = (int)all.size();
      wnp::ex(p).hpair_count[kd] = (int)hpairs[kd].size();
      all.insert(all.end(), hpairs[kd].begin(), hpairs[kd].end());
    }
    if (!all.empty()) {
      WN_HIP_CHECK(hipMalloc((void**)&wnp::ex(p).d_pairs, all.size() * sizeof(WnWgPair)));
      WN_HIP_CHECK(hipMemcpy(wnp::ex(p).d_pairs, all.data(), all.size() * sizeof(WnWgPair), hipMemcpyHostToDevice));
    }
  }
  wnp::ex(p).jobs_layerk = layerk; wnp::ex(p).jobs_pairk = pairk; wnp::ex(p).jobs_mfused = mfused; wnp::ex(p).jobs_mtr = mtr; wnp::ex(p).jobs_deep16 = d16; wnp::ex(p).jobs_headpairs = headpairs; wnp::ex(p).jobs_inconvk = inconvk;
  wnp::ex(p).jobs_fold = fold;
  wnp::ex(p).njobs = (int)jobs.size(); wnp::ex(p).ncov = (int)cov.size();
  wnp::ex(p).jobs_B = B; wnp::ex(p).jobs_T = T; wnp::ex(p).jobs_splits = L.bsplits; wnp::ex(p).jobs_drop = wnp::ex(p).drop_rate > 0.f;
  wnp::ex(p).jobs_skipk = skipk;
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// One call of wn_train_fwd_bwd: the caller's buffers, the workspace layout of (B, T) and the running max-abs slots of the
// gradient tensors; the phases of the step are its member functions, in launch order.
// ------------------------------------------------------------------------------------------
struct TrainCall {
  wn_plan* p; const float* params; const float* x_full; const float* cond;
  int B, T, global_batch, n_replicas;
  float* grads; float* loss_out; float* pred_out; float* ws; hipStream_t s;
  WsLayout L;
  int64_t rows;
  float* inputs;
  const float* fragbase; float* slab;
  // running max-abs scalars of the gradient tensors (operand scaling of the split-precision GEMMs): GF[i] | g_skipsum | GU[b] | GH[b] | GP[b][i]
  float* am; int nf; float* am_gskip;
  const float* mlast = nullptr;      // the mapped condition (input of every conv_cond)
  bool fold = false, cond_batched = false;
  float* am_GF(int i) const { return am + i; }
  float* am_GU(int b) const { return am + nf + 1 + b; }
  float* am_GH(int b) const { return am + nf + 1 + p->N + b; }
  float* am_GP(int b, int i) const { return am + nf + 1 + p->N + (p->N + 1) + b * (p->LPB - 1) + i; }

  // ---- forward + loss (+ the armed step sample, the L2 loss term, the range flag): src/model.py:319-334 ----
  int forward_and_loss() {
    int rc;
    { const int rcs = shift_split(x_full, B, T, inputs, ws + L.yt, s); if (rcs) return rcs; }
    if (wnp::ex(p).phase_on) (void)hipEventRecord(wnp::ex(p).phase_ev[0], s);
    // (zeroed before the forward pass: a fused loss epilogue publishes the max-abs of d loss / d logits from there)
    WN_HIP_CHECK(hipMemsetAsync(am, 0, L.n_absmax * sizeof(float), s));
    // 256-class categorical head: the loss rides in the head's last conv (LossFuse) unless the caller wants the probabilities
    LossFuse lf;
    memset(&lf, 0, sizeof(lf));
    const bool fuse = !pred_out && loss_fusable(p, rows);
    if (fuse) {
      rc = wn_launch_quantize(ws + L.yt, reinterpret_cast<int32_t*>(ws + L.target), rows, p->c.bits, s);
      if (rc) return rc;
      lf.target = reinterpret_cast<const int32_t*>(ws + L.target);
      lf.gscale = 1.0f / (float)global_batch;          // compute_average_loss, src/model.py:328-329
      lf.loss_rows = ws + L.loss_rows; lf.g_logits = ws + L.GF.back(); lf.absmax_out = am_GF(nf - 1);
      if (wnp::ex(p).step_sample && !wnp::ex(p).step_sample_det) {     // the armed sample_waveform(pred) draw of the step (src/model.py:338)
        lf.sample_out = wnp::ex(p).step_sample; lf.inv_lv = 1.0f / (float)(1 << (p->c.bits - 1));
        lf.seed = wnp::ex(p).step_sample_seed; lf.offset = wnp::ex(p).step_sample_off;
      }
    }
    rc = forward_core(p, params, inputs, true, cond, B, T, true, ws, L, s, nullptr, fuse ? &lf : nullptr);
    if (rc) return rc;
    if (lf.done && lf.sample_out) wnp::ex(p).step_sample = nullptr;     // drawn
    if (wnp::ex(p).phase_on) (void)hipEventRecord(wnp::ex(p).phase_ev[1], s);
    rc = loss_stage(p, B, T, global_batch, true, ws, L, loss_out, am_GF(nf - 1), s, lf.done);
    if (rc) return rc;
    if (pred_out) {
      if (p->c.head == WN_HEAD_CATEGORICAL) rc = wn_launch_softmax(ws + L.logits, pred_out, rows, p->Cout, s);
      else rc = hipMemcpyAsync(pred_out, ws + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess ? WN_OK : WN_E_HIP;
      if (rc) return rc;
    }
    if (wnp::ex(p).step_sample) {
      // sample_waveform(pred) of this step (src/model.py:338) drawn from the logits while they are still hot:
      // no (rows, C) probability tensor is written or re-read
      float* so = wnp::ex(p).step_sample;
      wnp::ex(p).step_sample = nullptr;
      if (p->c.head == WN_HEAD_CATEGORICAL) {
        if (wnp::ex(p).step_sample_det) {
          wn_set_error("step sample: deterministic categorical draws go through wn_sample_waveform");
          return WN_E_UNSUPPORTED;
        }
        rc = wn_launch_sample_rand_cat_logits(ws + L.logits, rows, p->Cout, p->c.bits, wnp::ex(p).step_sample_seed, wnp::ex(p).step_sample_off, so, s);
      } else {
        // mixture heads: the model output IS the logits tensor
        if (wnp::ex(p).step_sample_det) rc = wn_launch_sample_det(ws + L.logits, rows, p->Cout, p->c.num_mixtures, p->c.bits, so, s);
        else rc = wn_launch_sample_rand(ws + L.logits, rows, p->Cout, p->c.num_mixtures, p->c.bits, p->c.head, wnp::ex(p).step_sample_seed, wnp::ex(p).step_sample_off, so, s);
      }
      if (rc) return rc;
    }
    // the step's other two scalars are complete here too: the L2 regulariser's loss term (src/model.py:331-334) and the
    // range flag of the forward pass
    if (p->c.l2_reg_factor > 0.f) {
      float* norms = ws + L.loss_rows;   // free by now
      rc = wn_launch_sumsq(params, p->d_kdesc, (int)p->kdesc.size(), norms, s);
      if (rc) return rc;
      rc = wn_launch_sum(norms, (int64_t)p->kdesc.size(), p->c.l2_reg_factor / (float)n_replicas, loss_out + 1, ws + L.sum_scratch, s);
      if (rc) return rc;
    } else {
      rc = wn_launch_fill(loss_out + 1, 0.f, 1, s);
      if (rc) return rc;
    }
    // (with dropout the split kernels read H * mask / (1 - rate) while only H is published: compare against limit * (1 - rate))
    rc = wn_launch_guard_flag(ws + L.fwd_absmax, WN_RANGE_LIMIT * (wnp::ex(p).drop_rate > 0.f ? 1.f - wnp::ex(p).drop_rate : 1.f),
                              wn_debug_get(1) != 1, loss_out + 2, s);
    if (rc) return rc;
    if (wnp::ex(p).phase_on) (void)hipEventRecord(wnp::ex(p).phase_ev[2], s);
    return WN_OK;
  }

  // conv_cond of ONE block on the time-invariant mapped condition: dW_c = m^T dcb, db_c = sum_b dcb, g_m += dcb W_c^T
  int cond_block_bwd(const BlockInfo& bi) {
    const ConvInfo& c = bi.conv_cond;
    int r = wgrad(mlast, p->Cc, p->Cc, 0, ws + L.dcb, 2 * p->D, 2 * p->D, 1, B, grads + p->tensors[c.kernel_t].off,
                  grads + p->tensors[c.bias_t].off, nullptr, slab, s);
    if (r) return r;
    return Gemm(1, B, p->Cc, ceil32(p->Cc)).seg(ws + L.dcb, 2 * p->D, 2 * p->D, 0, fragbase + c.fragB)
        .addc(ws + L.g_m0, p->Cc).run(ws + L.g_m0, p->Cc, s);
  }

  // ---- head, last conv first: d loss / d logits (written by the loss stage into GF.back()) down to the skip sum ----
  int head_backward() {
    int rc;
    float* head_out = p->c.use_skip ? ws + L.g_skipsum : ws + L.GH[p->N];
    for (int i = (int)p->finals.size() - 1; i >= (fold ? 1 : 0); --i) {
      const ConvInfo& c = p->finals[i];
      float* dst = (i == 0) ? head_out : ws + L.GF[i - 1];
      // 128 / 256 input channels: the streamed kernel's second form in its backward-data instantiation
      if (c.frag16B >= 0 && wn_debug_get(1) != 1 && wn_gemm_planes16s_supported(c.cin, c.cout, 1, c.cout, c.cin) &&
          (int64_t)rows * c.cout * 4 < ((int64_t)1 << 32)) {
        WnGemmPlanesArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.z = ws + L.GF[i]; ga.ld = c.cout; ga.plane_k = c.cout; ga.nplanes = 1;
        ga.w16 = fragbase + c.frag16B; ga.act = p->c.activation;
        ga.y = dst; ga.ldy = c.cin; ga.N = c.cin; ga.B = B; ga.T = T;
        ga.bwd = 1; ga.absmax_in = am_GF(i);
        ga.absmax_out = i > 0 ? am_GF(i - 1) : (p->c.use_skip ? am_gskip : am_GH(p->N));
        if (i > 0) { ga.aux = ws + L.HA[i - 1]; ga.ld_aux = c.cin; }
        rc = wn_launch_gemm_planes16s(ga, s);
        if (rc) return rc;
        continue;
      }
      Gemm gm(B, T, c.cin, ceil32(c.cin));
      gm.seg(ws + L.GF[i], c.cout, c.cout, 0, fragbase + c.fragB);
      if (i > 0) gm.dact(ws + L.HA[i - 1], c.cin, p->c.activation);
      if (c.frag16B >= 0)
        gm.w16(fragbase + c.frag16B).absmax(am_GF(i), nullptr, i > 0 ? am_GF(i - 1) : (p->c.use_skip ? am_gskip : am_GH(p->N)));
      rc = gm.run(dst, c.cin, s);
      if (rc) return rc;
    }
    return WN_OK;
  }

  // ---- residual blocks, last to first: data gradients only, every g_u / g_x is kept for the deferred weight gradients ----
  int chain_backward() {
    int rc;
    // folded skip path: the gradient of the skip sum is never formed; the blocks contract dL/da = GF[0] with V(b)
    const float* g_skip = (p->c.use_skip && !fold) ? ws + L.g_skipsum : nullptr;
    if (p->c.use_skip) {
      rc = wn_launch_fill(ws + L.GH[p->N], 0.f, rows * p->R, s);   // nothing flows into the last block output
      if (rc) return rc;
    }
    // Two products per launch (wn_bwd_pair.hip): g_x(b+1) and, from it in registers, g_u(b).  The chain is then
    //   g_u(N-1) | { g_x(b+1), g_u(b) } for b = N-2 .. 0 | g_x(0)   = N + 1 launches instead of 2 N.
    const bool pairk = fold && p->N >= 2 && wnp::ex(p).drop_rate == 0.f && p->c.use_residual && (p->c.cond_inputs == 0 || cond_batched) &&
                       (wn_bwd_pair_supported(p->R, p->D, p->KS, p->fold_F0) || wn_bwd_s128_supported(p->R, p->D, p->KS, p->fold_F0)) &&
                       p->Dp == p->D &&
                       // (the streamed R = 128 pair kernel indexes with 32-bit byte offsets: the two-launch chain takes over beyond)
                       (p->R != 128 || (int64_t)rows * 2 * p->D * 4 < ((int64_t)1 << 32));
    for (int b = p->N - 1; b >= 0; --b) {
      BlockPtrs k = block_ptrs(p, b, params, fragbase, B, T);
      deep16_ptrs(p, b, fragbase, k);
      const BlockInfo& bi = p->blocks[b];
      if (pairk && b < p->N - 1) {
        const BlockPtrs k1 = block_ptrs(p, b + 1, params, fragbase, B, T);
        WnBwdPairArgs a;
        memset(&a, 0, sizeof(a));
        a.gu_in = ws + L.GU[b + 1]; a.gx_res = ws + L.GH[b + 2]; a.gf = ws + L.GF[0];
        a.ag = ws + L.AG[b]; a.z = ws + L.Z + (int64_t)b * rows * p->Dp; a.ldz = p->Dp;
        a.gx_out = ws + L.GH[b + 1]; a.gu_out = ws + L.GU[b];
        a.wx16 = k1.G16x; a.wu16 = k.G16uf;
        a.am_gu_in = am_GU(b + 1); a.am_gf = am_GF(0); a.am_gx = am_GH(b + 1); a.am_gu = am_GU(b);
        a.B = B; a.T = T; a.dil = k1.dil[0];
        if (!a.wx16 || !a.wu16) { wn_set_error("bwd_pair: weight images missing"); return WN_E_UNSUPPORTED; }
        rc = p->R == 128 ? wn_launch_bwd_s128(a, s) : wn_launch_bwd_pair(a, s);
        if (rc) return rc;
        if (b == 0) {
          // g_x(0): the gradient at the first block's input (only the input conv's weight gradients need it)
          Gemm gm(B, T, p->R, ceil32(p->R));
          for (int t = 0; t < p->KS; ++t)
            gm.seg(ws + L.GU[0], 2 * p->D, 2 * p->D, -(p->KS - 1 - t) * k.dil[0], k.Bd[0] + t * k.Bd_stride[0]);
          if (p->c.use_residual) gm.addc(ws + L.GH[1], p->R);
          gm.w16(k.G16x).absmax(am_GU(0), nullptr, am_GH(0));
          rc = gm.run(ws + L.GH[0], p->R, s);
          if (rc) return rc;
        }
        continue;
      }
      BlockBufs f;
      memset(&f, 0, sizeof(f));
      f.x = (wnp::ex(p).drop_rate > 0.f) ? ws + L.XD[b] : ws + L.H[b];
      for (int i = 0; i + 1 < p->LPB; ++i) f.P[i] = ws + L.P[b][i];
      f.AG = ws + L.AG[b];
      f.Z = ws + L.Z + (int64_t)b * rows * p->Dp; f.ldz = p->Dp;
      BlockGrads bg;
      memset(&bg, 0, sizeof(bg));
      bg.defer = true;
      for (int i = 0; i + 1 < p->LPB; ++i) bg.g_pi[i] = ws + L.GP[b][i];
      if (deep16(p))
        for (int i = 0; i + 1 < p->LPB; ++i) bg.am_gp[i] = am_GP(b, i);
      if (wnp::ex(p).drop_rate > 0.f) {
        bg.drop_rate = wnp::ex(p).drop_rate; bg.drop_key = wn_dropout_key(wnp::ex(p).drop_seed, b, wnp::ex(p).drop_step); bg.g_xd = ws + L.gxd;
      }
      // the last block's output gradient is identically zero when the head reads the skip sum
      // (with the skip head nothing flows into the last block's output: GH[N] was zero-filled above.  It is
      //  still passed as a gradient -- unless S == 0, where g_o is assembled from g_skip alone -- so that
      //  the last block runs the same split-precision kernels as the others instead of the fp32 fallback
      //  for the one-segment product)
      bg.g_xout = (p->c.use_skip && b == p->N - 1 && p->S == 0) ? nullptr : ws + L.GH[b + 1];
      bg.g_skip = g_skip;
      bg.g_o_tmp = p->S == 0 ? ws + L.GO[b] : nullptr;
      bg.g_u = ws + L.GU[b];
      bg.g_x = pairk ? nullptr : ws + L.GH[b];     // (pairs: the next launch forms g_x of this block)
      bg.dcb = (bi.has_cond && !cond_batched) ? ws + L.dcb : nullptr;
      bg.slab = slab;
      bg.am_gxout = bg.g_xout ? am_GH(b + 1) : nullptr;
      bg.am_gskip = g_skip ? am_gskip : nullptr;
      bg.am_gu = am_GU(b); bg.am_gx = am_GH(b);
      if (fold) { bg.g_fold = ws + L.GF[0]; bg.fold_F0 = p->fold_F0; bg.am_gfold = am_GF(0); }
      rc = block_backward(k, f, bg, s);
      if (rc) return rc;
      if (p->S == 0 && bg.g_xout == nullptr && g_skip) {
        // g_o == g_skip for this block: the job table reads GO[b]
        WN_HIP_CHECK(hipMemcpyAsync(ws + L.GO[b], g_skip, rows * p->R * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      if (bi.has_cond && !cond_batched) {
        rc = cond_block_bwd(bi);
        if (rc) return rc;
      }
    }
    return WN_OK;
  }

  // folded skip path: M = Z^T dL/da -> dW_s, db_s of every block and dW_f0, db_f0 (three small weight-space products)
  int fold_weight_gradients() {
    int rc = WN_OK;
    // M = Z^T dL/da (N*D x F0) and colsum(dL/da) into their own slab, reduced, then the three small products
    const int F0 = p->fold_F0;
    const int64_t pm = (int64_t)p->N * p->D * F0 + F0;
    if (wnp::ex(p).jobs_mtr != 0)
      rc = wn_launch_wgrad_tr(wnp::ex(p).jobs_mtr, wnp::ex(p).d_pairs + wnp::ex(p).pair_first[0], wnp::ex(p).pair_count[0], ws, ws + L.mslab, pm, B, T,
                              L.bsplits, s);
    else if (!wnp::ex(p).jobs_mfused)
    rc = wn_launch_wgrad_skip(ws + L.Z, p->Dp, ws + L.GF[0], F0, rows, p->N * p->D, F0, p->D, B * L.bsplits, ws + L.mslab, pm,
                              0, (int64_t)p->D * F0, (int64_t)p->N * p->D * F0, 0, 1, am_GF(0), s);
    if (rc) return rc;
    rc = wn_launch_reduce_table(ws + L.mslab, B * L.bsplits, pm, ws + L.mtot, p->d_cov_fold, 1, s, &p->h_cov_fold);
    if (rc) return rc;
    const BlockInfo& b0 = p->blocks[0];
    const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.kernel_t].off - p->tensors[b0.conv_skip.kernel_t].off : 0;
    const int64_t bst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.bias_t].off - p->tensors[b0.conv_skip.bias_t].off : 0;
    // Y = [M; colsum] W_f0^T -> dW_s of every block and db_s;  dW_f0 = [W_s(all); sum b_s]^T [M; colsum];  db_f0 = colsum
    const ConvInfo& c0 = p->finals[0];
    const int nd1 = p->N * p->D + 1;
    const float* wf0 = params + p->tensors[c0.kernel_t].off;          // (1, S, F0): W_f0[s][n]
    rc = wn_launch_sgemm_small(ws + L.mtot, F0, 1, wf0, 1, F0, ws + L.ytmp, p->S, nd1, p->S, F0, s);          // B[k = n][j = s]
    if (rc) return rc;
    // (a long-K product with a small output: the rows-contraction kernel splits K over workgroups)
    // split K in chunks of 128 on the small-product kernel, partial results in the (idle) slab, summed in chunk order
    // (wn_wgrad_kernel when the slab is too small for them)
    const int nzk = (nd1 + 127) / 128;
    if ((int64_t)nzk * p->S * F0 <= L.slab_floats) {
      rc = wn_launch_sgemm_small_batched(ws + L.wsall, 1, p->S, (int64_t)128 * p->S, ws + L.mtot, F0, 1, (int64_t)128 * F0, slab, F0,
                                         (int64_t)p->S * F0, p->S, F0, nd1, nzk, nullptr, 0, s, 128);
      if (rc) return rc;
      WnVecSumArgs v;
      v.base = slab; v.off0 = 0; v.stride = (int64_t)p->S * F0; v.count = nzk; v.len = p->S * F0;
      v.out = grads + p->tensors[c0.kernel_t].off;
      rc = wn_launch_vecsum(v, s);
    } else
    rc = wgrad(ws + L.wsall, p->S, p->S, 0, ws + L.mtot, F0, F0, 1, nd1, grads + p->tensors[c0.kernel_t].off, nullptr, nullptr,
               slab, s);
    if (rc) return rc;
    rc = wn_launch_skip_scatter(ws + L.ytmp, ws + L.mtot + (int64_t)p->N * p->D * F0, p->tensors[b0.conv_skip.kernel_t].off, wst,
                                p->tensors[b0.conv_skip.bias_t].off, bst, p->tensors[c0.bias_t].off, p->N, p->D, p->S, F0, grads, s);
    if (rc) return rc;
    return WN_OK;
  }

  // global conditioning of all blocks as one layer: per-utterance sums of d u out of the weight-gradient slab, g_m, dW_c, db_c
  int cond_weight_gradients() {
    int rc;
    const int D2 = 2 * p->D;
    const BlockInfo& b0 = p->blocks[0];
    const int64_t dst = p->N > 1 ? p->tensors[p->blocks[1].dil.back().bias_t].off - p->tensors[b0.dil.back().bias_t].off : 0;
    const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_cond.kernel_t].off - p->tensors[b0.conv_cond.kernel_t].off : 0;
    const int64_t bst = p->N > 1 ? p->tensors[p->blocks[1].conv_cond.bias_t].off - p->tensors[b0.conv_cond.bias_t].off : 0;
    rc = wn_launch_cond_gather(ws + L.bslab, p->nparams, L.bsplits, p->tensors[b0.dil.back().bias_t].off, dst, B, p->N, D2,
                               ws + L.cbt, s);
    if (rc) return rc;
    if (cond_small(p) && (int64_t)p->N * B * p->Cc <= L.slab_floats) {
      // g_m = sum_z dcb_z W_c(z)^T: one product per block into the (idle) slab, then their sum
      rc = wn_launch_sgemm_small_batched(ws + L.cbt, p->N * D2, 1, D2, params + p->tensors[b0.conv_cond.kernel_t].off, 1, D2, wst,
                                         slab, p->Cc, (int64_t)B * p->Cc, B, p->Cc, D2, p->N, nullptr, 0, s);
      if (rc) return rc;
      WnVecSumArgs v;
      v.base = slab; v.off0 = 0; v.stride = (int64_t)B * p->Cc; v.count = p->N; v.len = B * p->Cc; v.out = ws + L.g_m0;
      rc = wn_launch_vecsum(v, s);
    } else
    rc = Gemm(1, B, p->Cc, ceil32(p->Cc)).seg(ws + L.cbt, p->N * D2, p->N * D2, 0, fragbase + p->frag_condB).run(ws + L.g_m0, p->Cc, s);
    if (rc) return rc;
    rc = wn_launch_cond_wgrad(mlast, ws + L.cbt, B, p->Cc, p->N, D2, grads, p->tensors[b0.conv_cond.kernel_t].off, wst,
                              p->tensors[b0.conv_cond.bias_t].off, bst, s);
    if (rc) return rc;
    return WN_OK;
  }

  // ---- every weight gradient of the step: per-block kernels on the caller's stream, the low-occupancy leftovers (input
  //      conv, head) beside them on a side stream, then the slab reductions into the flat gradient ----
  int weight_gradients() {
    int rc;
    if (wnp::ex(p).phase_on) (void)hipEventRecord(wnp::ex(p).phase_ev[3], s);      // backward-data chain done
    // the generic jobs left over (input conv, head) are few single-wave jobs: they run beside the
    // per-block and skip kernels on a side stream (disjoint slab regions), joined before the reduce.
    // knob 9 = 1 keeps everything on the caller's stream (A/B of the overlap).
    const bool fork = (wnp::ex(p).jobs_layerk || wnp::ex(p).jobs_pairk) && wn_debug_get(9) != 1;
    if (fork && !wnp::ex(p).side) {
      WN_HIP_CHECK(hipStreamCreateWithFlags(&wnp::ex(p).side, hipStreamNonBlocking));
      WN_HIP_CHECK(hipEventCreateWithFlags(&wnp::ex(p).ev_fork, hipEventDisableTiming));
      WN_HIP_CHECK(hipEventCreateWithFlags(&wnp::ex(p).ev_join, hipEventDisableTiming));
    }
    // Whatever happens after the fork, the caller's stream must not run ahead of the side stream's kernels (they
    // read and write the workspace and the gradient slab): an early error return joins through this guard.
    struct SideJoin {
      wn_plan* p; hipStream_t s; bool armed;
      ~SideJoin() {
        if (!armed) return;
        if (hipEventRecord(wnp::ex(p).ev_join, wnp::ex(p).side) != hipSuccess || hipStreamWaitEvent(s, wnp::ex(p).ev_join, 0) != hipSuccess)
          (void)hipStreamSynchronize(wnp::ex(p).side);
      }
    } side_join{p, s, false};
    if (fork) {
      WN_HIP_CHECK(hipEventRecord(wnp::ex(p).ev_fork, s));
      WN_HIP_CHECK(hipStreamWaitEvent(wnp::ex(p).side, wnp::ex(p).ev_fork, 0));
      side_join.armed = true;
    }
    if (wnp::ex(p).jobs_inconvk) {
      rc = wn_launch_inconv_wgrad(inputs, ws + L.GH[0], B, T, p->R, p->KS, L.isplits, ws + L.islab, (int64_t)(p->KS + 1) * p->R,
                                  0, (int64_t)p->KS * p->R, fork ? wnp::ex(p).side : s);
      if (rc) return rc;
    }
    const bool head_own = L.hsplits > 0 && (wnp::ex(p).head_first < wnp::ex(p).njobs || wnp::ex(p).jobs_headpairs);
    rc = wn_launch_wgrad_batched(wnp::ex(p).d_jobs, head_own ? wnp::ex(p).head_first : wnp::ex(p).njobs, ws, ws + L.bslab, p->nparams, B, T, L.bsplits,
                                 fork ? wnp::ex(p).side : s, p->LPB > 1 && !wnp::ex(p).jobs_deep16);
    if (rc) return rc;
    if (head_own) {
      // job and coverage offsets are offsets into the flat parameter buffer: the compact slab is addressed
      // through a base shifted by -head_base with the head span as its row pitch
      if (wnp::ex(p).head_first < wnp::ex(p).njobs) {
        rc = wn_launch_wgrad_batched(wnp::ex(p).d_jobs + wnp::ex(p).head_first, wnp::ex(p).njobs - wnp::ex(p).head_first, ws, ws + L.hslab - L.head_base,
                                     L.head_span, B, T, L.hsplits, fork ? wnp::ex(p).side : s);
        if (rc) return rc;
      }
      if (wnp::ex(p).jobs_headpairs)
        for (int kd = 1; kd <= 5; ++kd)
          if (wnp::ex(p).hpair_count[kd] > 0) {
            // staged kinds 1 (128 x 256), 3 (256 x 128), 5 (256 x 256 halves) have transposed-read forms (3, 4, 5)
            const int trk = kd == 1 ? 3 : (kd == 3 ? 4 : (kd == 5 ? 5 : (kd == 2 ? 2 : 0)));
            if (trk != 0 && wnp::ex(p).jobs_pairk)                 // (with 64-channel blocks the staged head jobs are faster beside the side stream's neighbours)
              rc = wn_launch_wgrad_tr(trk, wnp::ex(p).d_pairs + wnp::ex(p).hpair_first[kd], wnp::ex(p).hpair_count[kd], ws, ws + L.hslab - L.head_base,
                                      L.head_span, B, T, L.hsplits, fork ? wnp::ex(p).side : s);
            else
            rc = wn_launch_wgrad_pairs(kd, wnp::ex(p).d_pairs + wnp::ex(p).hpair_first[kd], wnp::ex(p).hpair_count[kd], ws, ws + L.hslab - L.head_base,
                                       L.head_span, B, T, L.hsplits, fork ? wnp::ex(p).side : s);
            if (rc) return rc;
          }
    }
    if (fork) WN_HIP_CHECK(hipEventRecord(wnp::ex(p).ev_join, wnp::ex(p).side));
    for (int kd = 1; kd <= 2; ++kd)
      if (wnp::ex(p).jobs_pairk && wnp::ex(p).pair_count[kd] > 0) {
        // transposed-read kernels: both taps of dW_d in one job; dW_r (+ M)
        rc = wn_launch_wgrad_tr(kd == 2 && wnp::ex(p).jobs_mfused ? 6 : kd, wnp::ex(p).d_pairs + wnp::ex(p).pair_first[kd], wnp::ex(p).pair_count[kd], ws,
                                ws + L.bslab, p->nparams, B, T, L.bsplits, s, ws + L.mslab,
                                (int64_t)p->N * p->D * p->fold_F0 + p->fold_F0);
        if (rc) return rc;
      }
    if (wnp::ex(p).jobs_layerk) {
      rc = wn_launch_wgrad_layers(wnp::ex(p).d_wgl, p->N, p->R, ws, ws + L.bslab, p->nparams, B, T, L.bsplits, s);
      if (rc) return rc;
      rc = wn_launch_wgrad_layers(wnp::ex(p).d_wgli, wnp::ex(p).n_wgli, p->R, ws, ws + L.bslab, p->nparams, B, T, L.bsplits, s, 1);
      if (rc) return rc;
    }
    if (fold) {
      rc = fold_weight_gradients();
      if (rc) return rc;
    } else if (wnp::ex(p).jobs_skipk) {
      const BlockInfo& b0 = p->blocks[0];
      const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.kernel_t].off - p->tensors[b0.conv_skip.kernel_t].off : 0;
      const int64_t bst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.bias_t].off - p->tensors[b0.conv_skip.bias_t].off : 0;
      rc = wn_launch_wgrad_skip(ws + L.Z, p->Dp, ws + L.g_skipsum, p->S, rows, p->N * p->D, p->S, p->D,
                                B * L.bsplits, ws + L.bslab, p->nparams, p->tensors[b0.conv_skip.kernel_t].off, wst,
                                p->tensors[b0.conv_skip.bias_t].off, bst, p->N, am_gskip, s);
      if (rc) return rc;
    }
    if (fork) { WN_HIP_CHECK(hipStreamWaitEvent(s, wnp::ex(p).ev_join, 0)); side_join.armed = false; }
    if (cond_batched) {
      rc = cond_weight_gradients();
      if (rc) return rc;
    }
    // coverage entries 0, 1 are the input conv's kernel and bias: from their compact slab when the dedicated kernel ran
    const int cov0 = wnp::ex(p).jobs_inconvk ? 2 : 0;
    rc = wn_launch_reduce_table(ws + L.bslab, B * L.bsplits, p->nparams, grads, wnp::ex(p).d_cov + cov0,
                                (head_own ? wnp::ex(p).cov_head_first : wnp::ex(p).ncov) - cov0, s, wnp::ex(p).h_cov.data() + cov0);
    if (rc) return rc;
    if (wnp::ex(p).jobs_inconvk) {
      rc = wn_launch_reduce_table(ws + L.islab, B * L.isplits, (int64_t)(p->KS + 1) * p->R, grads, wnp::ex(p).d_cov, 2, s, wnp::ex(p).h_cov.data());
      if (rc) return rc;
    }
    if (head_own) {
      rc = wn_launch_reduce_table(ws + L.hslab - L.head_base, B * L.hsplits, L.head_span, grads, wnp::ex(p).d_cov + wnp::ex(p).cov_head_first,
                                  wnp::ex(p).ncov - wnp::ex(p).cov_head_first, s, wnp::ex(p).h_cov.data() + wnp::ex(p).cov_head_first);
      if (rc) return rc;
    }
    if (!p->c.use_skip && p->S > 0) {
      for (const BlockInfo& bi : p->blocks) {     // unused skip convs: zero gradients
        rc = wn_launch_fill(grads + p->tensors[bi.conv_skip.kernel_t].off, 0.f, p->tensors[bi.conv_skip.kernel_t].len, s);
        if (!rc) rc = wn_launch_fill(grads + p->tensors[bi.conv_skip.bias_t].off, 0.f, p->tensors[bi.conv_skip.bias_t].len, s);
        if (rc) return rc;
      }
    }
    return WN_OK;
  }

  int mapping_backward() {
    int rc;
    // ---- mapping Dense stack backward ----
    if (p->c.cond_inputs > 0) {
      const float* gm_cur = ws + L.g_m0;      // gradient w.r.t. post-activation output of the last Dense
      float* gm_other = ws + L.g_m1;
      for (int j = (int)p->mapping.size() - 1; j >= 0; --j) {
        const ConvInfo& c = p->mapping[j];
        const float* yin = (j == 0) ? cond : ws + L.M[j - 1];
        // pre-activation gradient g_pre = g * act'(M[j])  (tiny: B x width)
        rc = wn_launch_dact_mul(gm_cur, ws + L.M[j], gm_other, (int64_t)B * c.cout, p->c.mapping_activation, s);
        if (rc) return rc;
        if (cond_small(p)) {
          // dW = yin^T g_pre (cin x cout, contraction over the B utterances), db = column sums of g_pre
          rc = wn_launch_sgemm_small_batched(yin, 1, c.cin, 0, gm_other, c.cout, 1, 0, grads + p->tensors[c.kernel_t].off, c.cout, 0,
                                             c.cin, c.cout, B, 1, nullptr, 0, s);
          if (rc) return rc;
          WnVecSumArgs v;
          v.base = gm_other; v.off0 = 0; v.stride = c.cout; v.count = B; v.len = c.cout; v.out = grads + p->tensors[c.bias_t].off;
          rc = wn_launch_vecsum(v, s);
          if (rc) return rc;
          if (j > 0) {          // g_in = g_pre W^T
            float* dst = const_cast<float*>(gm_cur);
            rc = wn_launch_sgemm_small_batched(gm_other, c.cout, 1, 0, params + p->tensors[c.kernel_t].off, 1, c.cout, 0, dst, c.cin, 0,
                                               B, c.cin, c.cout, 1, nullptr, 0, s);
            if (rc) return rc;
          }
          continue;
        }
        rc = wgrad(yin, c.cin, c.cin, 0, gm_other, c.cout, c.cout, 1, B, grads + p->tensors[c.kernel_t].off,
                   grads + p->tensors[c.bias_t].off, nullptr, slab, s);
        if (rc) return rc;
        if (j > 0) {
          float* dst = const_cast<float*>(gm_cur);
          rc = Gemm(1, B, c.cin, ceil32(c.cin)).seg(gm_other, c.cout, c.cout, 0, fragbase + c.fragB).run(dst, c.cin, s);
          if (rc) return rc;
        }
      }
    }
    return WN_OK;
  }

  int backward() {
    int rc;
    fragbase = ws + L.frag;
    slab = ws + L.slab;
    if (p->c.cond_inputs > 0) {
      mlast = p->mapping.empty() ? cond : ws + L.M.back();
      rc = wn_launch_fill(ws + L.g_m0, 0.f, (int64_t)B * p->Cc, s);
      if (rc) return rc;
    }
    // conditioning of all blocks as one layer: the per-utterance sums of d u come out of the weight-gradient
    // slab afterwards instead of 2 column-sum launches + 3 tiny products per block
    cond_batched = p->frag_condB >= 0;
    fold = fold_ok(p);                             // (the forward pass of this call made the same decision)
    rc = ensure_jobs(p, L, B, T);
    if (rc) return rc;
    if ((rc = head_backward())) return rc;
    if ((rc = chain_backward())) return rc;
    if ((rc = weight_gradients())) return rc;
    if ((rc = mapping_backward())) return rc;
    // ---- L2 regulariser, src/model.py:331-334: its gradient (the loss term is formed with the loss, above) ----
    if (p->c.l2_reg_factor > 0.f) {
      rc = wn_launch_axpy_table(grads, params, p->d_kdesc, (int)p->kdesc.size(), 2.0f * p->c.l2_reg_factor / (float)n_replicas, s);
      if (rc) return rc;
    }
    if (wnp::ex(p).phase_on) {
      (void)hipEventRecord(wnp::ex(p).phase_ev[4], s);
    }

    return WN_OK;
  }
};

}  // namespace

extern "C" int wn_plan_set_train_phases(wn_plan* p, int32_t phases) {
  if (!p || phases < 1 || phases > 3) { wn_set_error("set_train_phases: 1 (forward + loss), 2 (backward), 3 (both)"); return WN_E_INVALID; }
  wnp::ex(p).train_phases = phases;
  return WN_OK;
}

extern "C" int wn_train_fwd_bwd(wn_plan* p, const float* params, const float* x_full, const float* cond,
                                int32_t B, int32_t T, int32_t global_batch, int32_t n_replicas, float* grads,
                                float* loss_out, float* pred_out, float* workspace, int64_t ws_floats,
                                void* stream) {
  if (!p || !params || !x_full || !workspace || !loss_out || !grads || B < 1 || T < 1) { wn_set_error("train_fwd_bwd: bad arguments"); return WN_E_INVALID; }
  TrainCall c;
  c.p = p; c.params = params; c.x_full = x_full; c.cond = cond; c.B = B; c.T = T;
  c.global_batch = global_batch > 0 ? global_batch : B;
  c.n_replicas = n_replicas > 0 ? n_replicas : 1;
  c.grads = grads; c.loss_out = loss_out; c.pred_out = pred_out; c.ws = workspace; c.s = (hipStream_t)stream;
  c.L = make_layout(p, B, T, true);
  if (ws_floats < c.L.total) { wn_set_error("train_fwd_bwd: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)c.L.total); return WN_E_INVALID; }
  c.rows = (int64_t)B * T;
  c.inputs = workspace + c.L.probs;
  c.am = workspace + c.L.absmax;
  c.nf = (int)p->finals.size();
  c.am_gskip = c.am + c.nf;
  // A caller may run the step as two calls (wn_plan_set_train_phases 1, then 2) and queue work of its own in between --
  // the Python mirror reads the loss and the metrics back from there, 4 ms before the step ends.  Everything the second
  // half needs lives in the workspace.
  const int phases = wnp::ex(p).train_phases;
  int rc = WN_OK;
  if (phases & 1) rc = c.forward_and_loss();
  if (!rc && (phases & 2)) rc = c.backward();
  return rc;
}

extern "C" int wn_adam_step_guarded(wn_plan* p, float* params, const float* grads, float* m, float* v, int64_t step,
                                    float lr, float beta1, float beta2, float eps, float clipnorm, float* scratch,
                                    const float* skip_flag, void* stream) {
  if (!p || !params || !grads || !m || !v || !scratch || step < 1) { wn_set_error("adam_step: bad arguments"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_device_tables(p);
  if (rc) return rc;
  const int n = (int)p->tdesc.size();
  if (clipnorm > 0.f) {
    rc = wn_launch_sumsq(grads, p->d_tdesc, n, scratch, s);
    if (rc) return rc;
  }
  const double alpha = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
  return wn_launch_adam(params, grads, m, v, p->d_tdesc, n, scratch, clipnorm, (float)alpha, beta1, beta2, eps, skip_flag, s);
}

extern "C" int wn_adam_step(wn_plan* p, float* params, const float* grads, float* m, float* v, int64_t step,
                            float lr, float beta1, float beta2, float eps, float clipnorm, float* scratch,
                            void* stream) {
  return wn_adam_step_guarded(p, params, grads, m, v, step, lr, beta1, beta2, eps, clipnorm, scratch, nullptr, stream);
}
