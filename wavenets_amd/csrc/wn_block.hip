// One residual block, forward and backward, on explicit pointers (WaveNetLayer.call, src/layers.py:178-224): shared by
// the model orchestration and by the standalone WaveNetLayer entry points at the end of this file.
#include "wn_plan_internal.h"

namespace wnp {

// dW (+ optional db, + optional per-utterance column sums) through slabs
int wgrad(const float* x, int ldx, int K, int shift, const float* g, int ldg, int N, int B, int T,
          float* dW, float* db, float* per_batch, float* slab, hipStream_t s) {
  WnWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = ldx; a.K = K; a.shift = shift; a.g = g; a.ldg = ldg; a.N = N; a.B = B; a.T = T;
  a.splits_per_b = wn_wgrad_choose_splits(B, T, K, N);
  const int nsplit = B * a.splits_per_b;
  a.slab = slab;
  a.slab_bias = (db || per_batch) ? slab + (int64_t)nsplit * K * N : nullptr;
  int rc = wn_launch_wgrad(a, s);
  if (rc) return rc;
  // per-utterance sums first: the reduces below fold the slabs in place
  if (per_batch) {
    rc = wn_launch_batch_reduce(a.slab_bias, B, a.splits_per_b, N, per_batch, s);
    if (rc) return rc;
  }
  if (dW) {
    WnReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.slab = slab; r.nsplit = nsplit; r.K = K; r.N = N; r.out = dW; r.seg_len = K; r.seg_stride = 0;
    r.accumulate = 0; r.replicate = 1; r.rep_stride = 0;
    rc = wn_launch_reduce(r, s);
    if (rc) return rc;
  }
  if (db) {
    WnReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.slab = a.slab_bias; r.nsplit = nsplit; r.K = 1; r.N = N; r.out = db; r.seg_len = 1; r.seg_stride = 0;
    r.accumulate = 0; r.replicate = 1; r.rep_stride = 0;
    rc = wn_launch_reduce(r, s);
    if (rc) return rc;
  }
  return rc;
}

int block_forward(const BlockPtrs& k, const BlockBufs& f, hipStream_t s) {
  const int64_t rows = (int64_t)k.B * k.T;
  const float* h = f.x;
  int hc = k.Cin;
  int rc;
  if (f.pre_done && k.depth > 1) { h = nullptr; hc = k.D; }
  for (int i = 0; i + 1 < k.depth && !f.pre_done; ++i) {
    const bool i16 = k.F16i[i] != nullptr;            // (training passes of deep stacks: split-precision, see deep16_ptrs)
    // 32 / 64 channels: the streamed kernel's second form with the taps as shifted planes (the rows GEMM for other widths)
    if (i16 && k.JTi[i] == 2 && k.KS <= 4 && wn_gemm_taps16s_supported(k.D, hc, k.KS, hc, k.D) &&
        (int64_t)rows * hc * 4 < ((int64_t)1 << 32)) {
      WnGemmPlanesArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.z = h; ga.plane_stride = 0; ga.ld = hc; ga.plane_k = hc; ga.nplanes = k.KS;
      ga.nshift = k.KS;
      for (int t = 0; t < k.KS; ++t) ga.shift[t] = (k.KS - 1 - t) * k.dil[i];
      ga.w16 = k.F16i[i]; ga.bias = k.bd[i]; ga.act = k.act;
      ga.y = f.P[i]; ga.ldy = k.D; ga.N = k.D; ga.B = k.B; ga.T = k.T;
      rc = wn_launch_gemm_planes16s(ga, s);
      if (rc) return rc;
      h = f.P[i]; hc = k.D;
      continue;
    }
    Gemm g(k.B, k.T, k.D, i16 ? k.JTi[i] : ceil32(k.D));
    for (int t = 0; t < k.KS; ++t) g.seg(h, hc, hc, (k.KS - 1 - t) * k.dil[i], i16 ? nullptr : k.Fd[i] + t * k.Fd_stride[i]);
    if (i16) g.w16(k.F16i[i]);
    rc = g.bias(k.bd[i]).act(k.act).run(f.P[i], k.D, s);
    if (rc) return rc;
    h = f.P[i]; hc = k.D;
  }
  const int li = k.depth - 1;
  // blocks too wide for LDS-resident weights (R = D = 128): one kernel that streams the fp16 hi|lo images through an LDS
  // ring and keeps u and z on chip (wn_layer16s.hip)
  if (k.F16n && k.F16r && k.depth == 1 && k.Cc == 0 && hc == k.R && k.Cin == k.R && f.ldz % 4 == 0 && wn_debug_get(1) != 1 &&
      wn_layer_fwd_s128_supported(k.R, k.D, k.KS) && (int64_t)rows * k.R * 4 < ((int64_t)1 << 32)) {
    WnLayerFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h; a.frag_d = k.F16n; a.frag_r = k.F16r;
    a.bias_d = k.bd[li]; a.bias_r = k.br; a.cb = k.cb;
    a.x_out = f.x_out; a.o_out = f.O; a.z_out = f.Z; a.ldz = f.ldz; a.ag_out = f.AG;
    a.res = f.res;                                      // (null: the residual is the conv input itself)
    a.xt[0] = f.xt[0]; a.xt[1] = f.xt[1]; a.xt[2] = nullptr;
    a.B = k.B; a.T = k.T; a.R = k.R; a.D = k.D; a.KS = k.KS; a.dilation = k.dil[li]; a.residual = k.residual;
    a.absmax_out = f.fwd_absmax;
    // one row per utterance (a queued-generation step): the whole block in one workgroup with every weight fragment
    // requested up front instead of the streamed pipeline (wn_gen128.hip, same arithmetic)
    if (k.T == 1 && f.xt[0] && f.xt[1] && !f.AG && wn_gen_block128_supported(k.R, k.D, k.KS))
      return wn_launch_gen_block128(a, s);
    return wn_launch_layer_fwd_s128(a, s);
  }
  // wider blocks (or other kernel sizes) as [gated conv + gate] -> [1x1 + residual], two split-precision contractions, ahead of
  // the exact-fp32 one-kernel forward
  if (k.F16g && k.F16r && k.Cc == 0 && !k.cb && !f.O && hc == k.R && f.ldz % 4 == 0 && wn_debug_get(1) != 1) {
    Gemm g(k.B, k.T, 2 * k.D, ceil32(2 * k.D));
    for (int t = 0; t < k.KS; ++t) {
      if (f.xt[t]) g.seg(f.xt[t], hc, hc, 0, nullptr);
      else g.seg(h, hc, hc, (k.KS - 1 - t) * k.dil[li], nullptr);
    }
    rc = g.bias(k.bd[li]).w16(k.F16g).gate_fwd(f.AG, k.D).run(f.Z, f.ldz, s);
    if (rc) return rc;
    Gemm r(k.B, k.T, k.R, ceil32(k.R));
    r.seg(f.Z, f.ldz, k.D, 0, k.Fr).bias(k.br).w16(k.F16r);
    if (k.residual) r.addc(f.res ? f.res : f.x, k.Cin);
    if (f.fwd_absmax) r.absmax_fwd(f.fwd_absmax);
    return r.run(f.x_out, k.R, s);
  }
  if (k.fused && k.Cc == 0 && hc == k.R) {
    WnLayerFwdArgs a;
    memset(&a, 0, sizeof(a));
    // knob 1 = 1 forces the exact-fp32 MFMA kernel
    const bool use16 = k.F16d && k.F16r && wn_debug_get(1) != 1;
    a.x = h; a.frag_d = use16 ? k.F16d : k.Fd[li]; a.frag_r = use16 ? k.F16r : k.Fr;
    a.bias_d = k.bd[li]; a.bias_r = k.br; a.cb = k.cb;
    a.x_out = f.x_out; a.o_out = f.O; a.z_out = f.Z; a.ldz = f.ldz; a.ag_out = f.AG;
    a.res = f.res ? f.res : ((k.depth > 1) ? f.x : nullptr);
    a.xt[0] = f.xt[0]; a.xt[1] = f.xt[1]; a.xt[2] = f.xt[2];
    a.B = k.B; a.T = k.T; a.R = k.R; a.D = k.D; a.KS = k.KS; a.dilation = k.dil[li]; a.residual = k.residual;
    a.absmax_out = f.fwd_absmax;
    return use16 ? wn_launch_layer_fwd_f16(a, s) : wn_launch_layer_fwd(a, s);
  }
  // composed path: u -> gate -> 1x1
  {
    Gemm g(k.B, k.T, 2 * k.D, ceil32(2 * k.D));
    for (int t = 0; t < k.KS; ++t) {
      if (f.xt[t]) g.seg(f.xt[t], hc, hc, 0, k.Fd[li] + t * k.Fd_stride[li]);
      else g.seg(h, hc, hc, (k.KS - 1 - t) * k.dil[li], k.Fd[li] + t * k.Fd_stride[li]);
    }
    if (k.Cc > 0) g.seg(k.cond, k.Cc, k.Cc, 0, k.Fc);
    g.bias(k.bd[li]);
    if (k.cb) g.rowbias(k.cb, 2 * k.D);
    rc = g.run(f.U, 2 * k.D, s);
    if (rc) return rc;
  }
  rc = wn_launch_gate(f.U, rows, k.D, f.AG, f.Z, f.ldz, s);
  if (rc) return rc;
  {
    Gemm g(k.B, k.T, k.R, ceil32(k.R));
    g.seg(f.Z, f.ldz, k.D, 0, k.Fr).bias(k.br);
    if (f.O) {
      rc = g.run(f.O, k.R, s);
      if (rc) return rc;
      if (k.residual) return wn_launch_add(f.O, f.res ? f.res : f.x, f.x_out, rows * k.R, s);
      return hipMemcpyAsync(f.x_out, f.O, rows * k.R * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess ? WN_OK : WN_E_HIP;
    }
    if (k.residual) g.addc(f.res ? f.res : f.x, k.Cin);
    return g.run(f.x_out, k.R, s);
  }
}

int block_backward(const BlockPtrs& k, const BlockBufs& f, const BlockGrads& g, hipStream_t s) {
  const int64_t rows = (int64_t)k.B * k.T;
  int rc;
  const int li = k.depth - 1;
  const float* hin_last = (k.depth > 1) ? f.P[li - 1] : f.x;
  const int hc_last = (k.depth > 1) ? k.D : k.Cin;
  // gradient w.r.t. the conv1 output o
  const float* g_o = g.g_xout;
  if (k.S == 0 && g.g_skip) {
    if (g.g_xout) {
      rc = wn_launch_add(g.g_xout, g.g_skip, g.g_o_tmp, rows * k.R, s);
      if (rc) return rc;
      g_o = g.g_o_tmp;
    } else {
      g_o = g.g_skip;
    }
  }
  // g_u = gate'( W_r g_o + W_s g_skip )
  {
    // (deep stacks in training: the [W_r | W_s] image may be padded to two row tiles; it is only set when it will be used)
    const bool full_u = (k.S > 0) ? (g_o && g.g_skip) : (g_o != nullptr);
    const bool pad_u = k.JTu > 0 && k.G16u && full_u && k.Cc == 0 && g.am_gu &&
                       ((g_o == g.g_xout) ? g.am_gxout : (g_o == g.g_skip ? g.am_gskip : g.am_gxout)) != nullptr;
    Gemm gm(k.B, k.T, k.D, pad_u ? k.JTu : ceil32(k.D));
    const bool use_fold = g.g_fold && g_o && k.G16uf && g.am_gu && g.am_gxout && g.am_gfold && k.Cc == 0;
    if (g_o) gm.seg(g_o, k.R, k.R, 0, (use_fold || pad_u) ? nullptr : k.Br_);
    if (use_fold) gm.seg(g.g_fold, g.fold_F0, g.fold_F0, 0, nullptr).w16(k.G16uf).absmax(g.am_gxout, g.am_gfold, g.am_gu);
    else if (k.S > 0 && g.g_skip) gm.seg(g.g_skip, k.S, k.S, 0, pad_u ? nullptr : k.Bs);
    if (use_fold) {
      rc = gm.gate_bwd(f.AG, k.D, f.Z, f.ldz).run(g.g_u, 2 * k.D, s);
    } else if (gm.a.nseg == 0) {
      rc = wn_launch_fill(g.g_u, 0.f, rows * 2 * k.D, s);
    } else {
      // the [W_r | W_s] image matches the segment list only when both (or, for S == 0, the single) operands exist
      const bool full = (k.S > 0) ? (g_o && g.g_skip) : true;
      if (k.G16u && full && k.Cc == 0 && (k.JTu == 0 || pad_u)) {
        const float* a0 = (g_o == g.g_xout) ? g.am_gxout : (g_o == g.g_skip ? g.am_gskip : g.am_gxout);
        const float* a1 = (k.S > 0 || g_o == g.g_o_tmp) ? g.am_gskip : nullptr;
        if (g.am_gu && a0) gm.w16(k.G16u).absmax(a0, a1, g.am_gu);
      }
      rc = gm.gate_bwd(f.AG, k.D, f.Z, f.ldz).run(g.g_u, 2 * k.D, s);
    }
    if (rc) return rc;
  }
  // conv1 / conv_skip weight gradients
  if (g.defer) {
    // batched later; only the per-utterance column sums of g_u are needed now (conditioning)
    if (g.dcb) {
      rc = wn_launch_colsum_per_batch(g.g_u, k.B, k.T, 2 * k.D, g.dcb, g.slab, s);
      if (rc) return rc;
    }
  } else if (g_o) {
    rc = wgrad(f.Z, f.ldz, k.D, 0, g_o, k.R, k.R, k.B, k.T, g.dWr, g.dbr, nullptr, g.slab, s);
  } else {
    rc = wn_launch_fill(g.dWr, 0.f, (int64_t)k.D * k.R, s);
    if (!rc) rc = wn_launch_fill(g.dbr, 0.f, k.R, s);
  }
  if (rc) return rc;
  if (!g.defer && k.S > 0 && g.dWs) {
    if (g.g_skip) {
      rc = wgrad(f.Z, f.ldz, k.D, 0, g.g_skip, k.S, k.S, k.B, k.T, g.dWs, g.dbs, nullptr, g.slab, s);
    } else {
      rc = wn_launch_fill(g.dWs, 0.f, (int64_t)k.D * k.S, s);
      if (!rc) rc = wn_launch_fill(g.dbs, 0.f, k.S, s);
    }
    if (rc) return rc;
  }
  // time-varying condition (standalone layer)
  if (k.Cc > 0) {
    rc = wgrad(k.cond, k.Cc, k.Cc, 0, g.g_u, 2 * k.D, 2 * k.D, k.B, k.T, g.dWc, g.dbc, nullptr, g.slab, s);
    if (rc) return rc;
    if (g.g_cond) {
      rc = Gemm(k.B, k.T, k.Cc, ceil32(k.Cc)).seg(g.g_u, 2 * k.D, 2 * k.D, 0, k.Bc).run(g.g_cond, k.Cc, s);
      if (rc) return rc;
    }
  }
  // dilated stack, last (gated) conv first
  const float* gcur = g.g_u;       // gradient w.r.t. the pre-activation output of conv i
  int gc = 2 * k.D;
  for (int i = li; i >= 0; --i) {
    const float* hin = (i > 0) ? f.P[i - 1] : f.x;
    const int hc = (i > 0) ? k.D : k.Cin;
    for (int t = 0; t < k.KS && !g.defer; ++t) {
      const bool last_tap = (t == k.KS - 1);
      rc = wgrad(hin, hc, hc, (k.KS - 1 - t) * k.dil[i], gcur, gc, gc, k.B, k.T,
                 g.dWd[i] + (int64_t)t * hc * gc, last_tap ? g.dbd[i] : nullptr,
                 (last_tap && i == li) ? g.dcb : nullptr, g.slab, s);
      if (rc) return rc;
    }
    const bool need_gx = (i > 0) || g.g_x;
    if (!need_gx) break;
    // deep stacks in training: split-precision product, operand scaled by the running max-abs of gcur, the result's
    // max-abs published for the next product and for the weight-gradient jobs
    const float* am_cur = (i == li) ? g.am_gu : g.am_gp[i];
    float* am_dst = (i > 0) ? g.am_gp[i - 1] : g.am_gx;
    const bool b16 = k.depth > 1 && k.G16i[i] && am_cur && am_dst;
    // inner convs (their output gradient gets act' folded in): the streamed kernel's second form, backward-data
    // instantiation with the taps as negatively shifted planes (the rows GEMM for other widths)
    if (b16 && i > 0 && k.JTb[i] == 2 && k.KS <= 4 && wn_gemm_taps16s_supported(hc, gc, k.KS, gc, hc) &&
        (int64_t)rows * gc * 4 < ((int64_t)1 << 32)) {
      float* dst = g.g_pi[i - 1] ? g.g_pi[i - 1] : g.g_p + (int64_t)((i & 1) ? 0 : rows * k.D);
      WnGemmPlanesArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.z = gcur; ga.plane_stride = 0; ga.ld = gc; ga.plane_k = gc; ga.nplanes = k.KS;
      ga.nshift = k.KS;
      for (int t = 0; t < k.KS; ++t) ga.shift[t] = -(k.KS - 1 - t) * k.dil[i];
      ga.w16 = k.G16i[i]; ga.act = k.act;
      ga.y = dst; ga.ldy = hc; ga.N = hc; ga.B = k.B; ga.T = k.T;
      ga.bwd = 1; ga.absmax_in = am_cur; ga.absmax_out = am_dst; ga.aux = f.P[i - 1]; ga.ld_aux = k.D;
      rc = wn_launch_gemm_planes16s(ga, s);
      if (rc) return rc;
      gcur = dst; gc = k.D;
      continue;
    }
    Gemm gm(k.B, k.T, hc, b16 ? k.JTb[i] : ceil32(hc));
    for (int t = 0; t < k.KS; ++t)
      gm.seg(gcur, gc, gc, -(k.KS - 1 - t) * k.dil[i], b16 ? nullptr : k.Bd[i] + t * k.Bd_stride[i]);
    if (i > 0) {
      // output is the gradient w.r.t. P[i-1] (post-activation) -> fold act' in
      float* dst = g.g_pi[i - 1] ? g.g_pi[i - 1] : g.g_p + (int64_t)((i & 1) ? 0 : rows * k.D);
      if (b16) gm.w16(k.G16i[i]).absmax(am_cur, nullptr, am_dst);
      rc = gm.dact(f.P[i - 1], k.D, k.act).run(dst, k.D, s);
      if (rc) return rc;
      gcur = dst; gc = k.D;
    } else {
      if (g.drop_rate > 0.f) {
        // conv-path gradient first, then the keep-mask, then the (unmasked) residual path
        if (k.G16x && k.depth == 1 && g.am_gu) gm.w16(k.G16x).absmax(g.am_gu, nullptr, nullptr);
        else if (b16) gm.w16(k.G16i[0]).absmax(am_cur, nullptr, nullptr);      // (the dropout kernel publishes g_x's max-abs)
        rc = gm.run(g.g_xd, hc, s);
        if (rc) return rc;
        rc = wn_launch_dropout(g.g_xd, (k.residual && g.g_xout) ? g.g_xout : nullptr, g.g_x, rows * hc, g.drop_rate,
                               g.drop_key, g.am_gx, s);
        if (rc) return rc;
        continue;
      }
      if (k.residual && g.g_xout) gm.addc(g.g_xout, k.R);
      if (k.G16x && k.depth == 1 && g.am_gu && g.am_gx) gm.w16(k.G16x).absmax(g.am_gu, nullptr, g.am_gx);
      else if (b16) gm.w16(k.G16i[0]).absmax(am_cur, nullptr, am_dst);
      rc = gm.run(g.g_x, hc, s);
      if (rc) return rc;
    }
  }
  (void)hin_last; (void)hc_last;
  return WN_OK;
}

}  // namespace wnp

using namespace wnp;

// ==========================================================================================
// standalone residual block: WaveNetLayer.call, src/layers.py:178-224
// ==========================================================================================
namespace {

struct LayerLayout {
  // parameter offsets (floats) inside the layer's flat parameter buffer (Keras order)
  int64_t Wd[16], bd[16], Wr, br, Ws, bs, Wc, bc, nparams;
  int cin[16], cout[16];
  // workspace
  int64_t Fd[16], Bd[16], Fd_stride[16], Bd_stride[16], Fr, Br, Bs, Fs, Fc, Bc;
  int64_t bias_u, U, O, g_u, g_o, g_p, slab, ws_total;
  int64_t F16d, F16r;   // fp16 split images (or -1)
  // saved
  int64_t sP[16], sAG, sZ, saved_total;
};

int layer_layout(const wn_layer_desc* d, int B, int T, LayerLayout& L) {
  if (!d || d->depth < 1 || d->depth > 16 || d->kernel_size < 2 || d->kernel_size > 3 || d->channels < 1) {
    wn_set_error("layer: bad descriptor"); return WN_E_INVALID;
  }
  const int KS = d->kernel_size, R = d->channels, D = d->dilation_channels > 0 ? d->dilation_channels : R;
  const int S = d->skip_channels, Cc = d->cond_channels;
  const int64_t rows = (int64_t)B * T;
  int64_t o = 0;
  int cin = d->in_channels > 0 ? d->in_channels : R;
  for (int i = 0; i < d->depth; ++i) {
    const int cout = (i == d->depth - 1) ? 2 * D : D;
    L.cin[i] = cin; L.cout[i] = cout;
    L.Wd[i] = o; o += (int64_t)KS * cin * cout;
    L.bd[i] = o; o += cout;
    cin = cout;
  }
  L.Wr = o; o += (int64_t)D * R; L.br = o; o += R;
  L.Ws = L.bs = L.Wc = L.bc = -1;
  if (S > 0) { L.Ws = o; o += (int64_t)D * S; L.bs = o; o += S; }
  if (Cc > 0) { L.Wc = o; o += (int64_t)Cc * 2 * D; L.bc = o; o += 2 * D; }
  L.nparams = o;
  Carver cv;
  for (int i = 0; i < d->depth; ++i) {
    L.Fd_stride[i] = (int64_t)wn_frag_floats(L.cout[i], L.cin[i]);
    L.Bd_stride[i] = (int64_t)wn_frag_floats(L.cin[i], L.cout[i]);
    L.Fd[i] = cv.take(KS * L.Fd_stride[i]);
    L.Bd[i] = cv.take(KS * L.Bd_stride[i]);
  }
  L.Fr = cv.take((int64_t)wn_frag_floats(R, D));
  L.Br = cv.take((int64_t)wn_frag_floats(D, R));
  L.Fs = cv.take(S > 0 ? (int64_t)wn_frag_floats(S, D) : 0);
  L.Bs = cv.take(S > 0 ? (int64_t)wn_frag_floats(D, S) : 0);
  L.Fc = cv.take(Cc > 0 ? (int64_t)wn_frag_floats(2 * D, Cc) : 0);
  L.Bc = cv.take(Cc > 0 ? (int64_t)wn_frag_floats(Cc, 2 * D) : 0);
  L.F16d = L.F16r = -1;
  if (d->depth == 1 && Cc == 0 && cin == 2 * D && (d->in_channels > 0 ? d->in_channels : R) == R &&
      ((wn_layer_fwd_supported(R, D, KS) && wn_layer_fwd_f16_supported(R, D, KS)) || wn_layer_fwd_s128_supported(R, D, KS))) {
    L.F16d = cv.take((int64_t)wn_frag16_floats(2 * D, KS * R));
    L.F16r = cv.take((int64_t)wn_frag16_floats(R, D));
  }
  L.bias_u = cv.take(2 * D);
  L.U = cv.take(rows * 2 * D);
  L.O = cv.take(rows * R);
  L.g_u = L.U;                     // backward reuses the u scratch
  L.g_o = L.O;
  L.g_p = cv.take(d->depth > 1 ? 2 * rows * D : 0);
  int64_t need = 0;
  if (B > 0 && T > 0) {
    for (int i = 0; i < d->depth; ++i) need = std::max(need, slab_need(B, T, L.cin[i], L.cout[i]));
    need = std::max(need, slab_need(B, T, D, R));
    if (S > 0) need = std::max(need, slab_need(B, T, D, S));
    if (Cc > 0) need = std::max(need, slab_need(B, T, Cc, 2 * D));
  }
  L.slab = cv.take(need);
  L.ws_total = cv.pos;
  Carver sv;
  for (int i = 0; i + 1 < d->depth; ++i) L.sP[i] = sv.take(rows * D);
  L.sAG = sv.take(rows * D);
  L.sZ = sv.take(rows * D);
  L.saved_total = sv.pos;
  return WN_OK;
}

int layer_prep(const wn_layer_desc* d, const LayerLayout& L, const float* params, float* ws, hipStream_t s) {
  const int KS = d->kernel_size, R = d->channels, D = d->dilation_channels > 0 ? d->dilation_channels : R;
  const int S = d->skip_channels, Cc = d->cond_channels;
  auto one = [&](int64_t src, int64_t dst, int I, int KK, int ld, int tr) {
    WnPrepDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.src_off = src; pd.dst_off = dst; pd.I = I; pd.KK = KK; pd.ld = ld; pd.transpose = tr;
    pd.JT = (I + 31) / 32;
    return wn_launch_prep_one(pd, params, ws, s);
  };
  int rc;
  for (int i = 0; i < d->depth; ++i)
    for (int t = 0; t < KS; ++t) {
      const int64_t src = L.Wd[i] + (int64_t)t * L.cin[i] * L.cout[i];
      if ((rc = one(src, L.Fd[i] + t * L.Fd_stride[i], L.cout[i], L.cin[i], L.cout[i], 1))) return rc;
      if ((rc = one(src, L.Bd[i] + t * L.Bd_stride[i], L.cin[i], L.cout[i], L.cout[i], 0))) return rc;
    }
  if ((rc = one(L.Wr, L.Fr, R, D, R, 1))) return rc;
  if ((rc = one(L.Wr, L.Br, D, R, R, 0))) return rc;
  if (L.F16d >= 0) {
    for (int t = 0; t < KS; ++t) {
      WnPrepDesc pd;
      memset(&pd, 0, sizeof(pd));
      pd.src_off = L.Wd[0] + (int64_t)t * R * 2 * D; pd.dst_off = L.F16d; pd.I = 2 * D; pd.KK = R; pd.ld = 2 * D;
      pd.transpose = 1; pd.q_off = t * (R / 16); pd.JT = (2 * D + 31) / 32; pd.kind = 1;
      if ((rc = wn_launch_prep_one(pd, params, ws, s))) return rc;
    }
    WnPrepDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.src_off = L.Wr; pd.dst_off = L.F16r; pd.I = R; pd.KK = D; pd.ld = R; pd.transpose = 1; pd.JT = (R + 31) / 32; pd.kind = 1;
    if ((rc = wn_launch_prep_one(pd, params, ws, s))) return rc;
  }
  if (S > 0) {
    if ((rc = one(L.Ws, L.Fs, S, D, S, 1))) return rc;
    if ((rc = one(L.Ws, L.Bs, D, S, S, 0))) return rc;
  }
  if (Cc > 0) {
    if ((rc = one(L.Wc, L.Fc, 2 * D, Cc, 2 * D, 1))) return rc;
    if ((rc = one(L.Wc, L.Bc, Cc, 2 * D, 2 * D, 0))) return rc;
    // u's bias = last dilated conv bias + conv_cond bias (src/layers.py:82-88,116-120,203-204)
    WnVecSumArgs v;
    v.base = params; v.off0 = L.bd[d->depth - 1]; v.stride = L.bc - L.bd[d->depth - 1]; v.count = 2; v.len = 2 * D;
    v.out = ws + L.bias_u;
    if ((rc = wn_launch_vecsum(v, s))) return rc;
  }
  return WN_OK;
}

void layer_ptrs(const wn_layer_desc* d, const LayerLayout& L, const float* params, const float* ws,
                const float* cond, int B, int T, BlockPtrs& k) {
  memset(&k, 0, sizeof(k));
  const int R = d->channels, D = d->dilation_channels > 0 ? d->dilation_channels : R;
  k.B = B; k.T = T; k.KS = d->kernel_size; k.R = R; k.D = D; k.S = d->skip_channels;
  k.Cin = d->in_channels > 0 ? d->in_channels : R; k.depth = d->depth; k.act = d->activation; k.residual = d->residual;
  for (int i = 0; i < d->depth; ++i) {
    k.dil[i] = d->dilations[i];
    k.Wd[i] = params + L.Wd[i]; k.bd[i] = params + L.bd[i];
    k.Fd[i] = ws + L.Fd[i]; k.Bd[i] = ws + L.Bd[i]; k.Fd_stride[i] = L.Fd_stride[i]; k.Bd_stride[i] = L.Bd_stride[i];
  }
  k.br = params + L.br; k.Fr = ws + L.Fr; k.Br_ = ws + L.Br;
  if (d->skip_channels > 0) { k.bs = params + L.bs; k.Fs = ws + L.Fs; k.Bs = ws + L.Bs; }
  k.Cc = d->cond_channels; k.cond = cond; k.cb = nullptr;
  if (k.Cc > 0) { k.Fc = ws + L.Fc; k.Bc = ws + L.Bc; k.bc = params + L.bc; k.bd[d->depth - 1] = ws + L.bias_u; }
  k.fused = wn_layer_fwd_supported(R, D, d->kernel_size) != 0;
  if (L.F16d >= 0) {
    // (the natural-order image serves the LDS-resident kernel or, for 128 channels, the streamed one)
    if (wn_layer_fwd_f16_supported(R, D, d->kernel_size)) k.F16d = ws + L.F16d; else k.F16n = ws + L.F16d;
    k.F16r = ws + L.F16r;
  }
}

}  // namespace

extern "C" int64_t wn_layer_param_count(const wn_layer_desc* d) {
  LayerLayout L;
  if (layer_layout(d, 0, 0, L)) return -1;
  return L.nparams;
}
extern "C" int64_t wn_layer_saved_floats(const wn_layer_desc* d, int32_t B, int32_t T) {
  LayerLayout L;
  if (layer_layout(d, B, T, L)) return -1;
  return L.saved_total;
}
extern "C" int64_t wn_layer_workspace_floats(const wn_layer_desc* d, int32_t B, int32_t T) {
  LayerLayout L;
  if (layer_layout(d, B, T, L)) return -1;
  return L.ws_total;
}

extern "C" int wn_layer_fwd(const wn_layer_desc* d, const float* params, const float* x, const float* cond,
                            int32_t B, int32_t T, float* x_out, float* skip_out, float* saved,
                            float* workspace, void* stream) {
  LayerLayout L;
  int rc = layer_layout(d, B, T, L);
  if (rc) return rc;
  if (!params || !x || !x_out || !workspace) { wn_set_error("layer_fwd: null pointer"); return WN_E_INVALID; }
  if (d->cond_channels > 0 && !cond) { wn_set_error("layer_fwd: condition tensor missing"); return WN_E_INVALID; }
  if (d->residual && (d->in_channels > 0 ? d->in_channels : d->channels) != d->channels) {
    wn_set_error("Residual connection must have the same shape as input"); return WN_E_INVALID;   // src/layers.py:161-162
  }
  hipStream_t s = (hipStream_t)stream;
  if ((rc = layer_prep(d, L, params, workspace, s))) return rc;
  BlockPtrs k;
  layer_ptrs(d, L, params, workspace, cond, B, T, k);
  const int D = k.D;
  const int64_t rows = (int64_t)B * T;
  BlockBufs f;
  memset(&f, 0, sizeof(f));
  f.x = x;
  // without a saved buffer the intermediates live in scratch carved after the u buffer
  float* sv = saved;
  for (int i = 0; i + 1 < d->depth; ++i) f.P[i] = sv ? sv + L.sP[i] : nullptr;
  f.U = workspace + L.U;
  f.AG = sv ? sv + L.sAG : nullptr;
  f.Z = sv ? sv + L.sZ : nullptr; f.ldz = D;
  if (!sv) { wn_set_error("layer_fwd: saved buffer is required (holds z and the stack activations)"); return WN_E_INVALID; }
  const bool skip_is_o = (d->skip_channels == 0);
  f.O = (skip_is_o && skip_out) ? skip_out : nullptr;
  f.x_out = x_out;
  if ((rc = block_forward(k, f, s))) return rc;
  if (!skip_is_o && skip_out) {
    rc = Gemm(B, T, d->skip_channels, ceil32(d->skip_channels)).seg(f.Z, D, D, 0, k.Fs).bias(k.bs).run(skip_out, d->skip_channels, s);
  }
  (void)rows;
  return rc;
}

extern "C" int wn_layer_bwd(const wn_layer_desc* d, const float* params, const float* x, const float* cond,
                            const float* saved, const float* g_x_out, const float* g_skip, int32_t B, int32_t T,
                            float* g_x, float* g_cond, float* g_params, float* workspace, void* stream) {
  LayerLayout L;
  int rc = layer_layout(d, B, T, L);
  if (rc) return rc;
  if (!params || !x || !saved || !g_params || !workspace) { wn_set_error("layer_bwd: null pointer"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  if ((rc = layer_prep(d, L, params, workspace, s))) return rc;
  BlockPtrs k;
  layer_ptrs(d, L, params, workspace, cond, B, T, k);
  BlockBufs f;
  memset(&f, 0, sizeof(f));
  f.x = x;
  float* sv = const_cast<float*>(saved);
  for (int i = 0; i + 1 < d->depth; ++i) f.P[i] = sv + L.sP[i];
  f.AG = sv + L.sAG; f.Z = sv + L.sZ; f.ldz = k.D;
  BlockGrads g;
  memset(&g, 0, sizeof(g));
  g.g_xout = g_x_out; g.g_skip = g_skip; g.g_o_tmp = workspace + L.g_o; g.g_u = workspace + L.g_u;
  g.g_p = workspace + L.g_p; g.g_x = g_x; g.g_cond = g_cond;
  for (int i = 0; i < d->depth; ++i) { g.dWd[i] = g_params + L.Wd[i]; g.dbd[i] = g_params + L.bd[i]; }
  g.dWr = g_params + L.Wr; g.dbr = g_params + L.br;
  if (d->skip_channels > 0) { g.dWs = g_params + L.Ws; g.dbs = g_params + L.bs; }
  if (d->cond_channels > 0) { g.dWc = g_params + L.Wc; g.dbc = g_params + L.bc; }
  g.slab = workspace + L.slab;
  return block_backward(k, f, g, s);
}
