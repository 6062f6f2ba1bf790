// Plan + orchestration of the WaveNet hot path behind the C-ABI of include/wn_hip.h.
//
// Host-side only (no kernels here): parameter layout in Keras creation order, the table of
// fragment-major weight images, workspace carving, and the launch sequences that restate
//   WaveNet.call          src/model.py:213-239
//   WaveNet.train_step    src/model.py:309-348 (gradient half)
//   WaveNetLayer.call     src/layers.py:178-224
// Nothing here allocates caller-visible memory; the two small device tables a plan owns
// (prep descriptors, tensor table) are immutable launch metadata uploaded on first use.
#include "wn_plan_internal.h"

namespace {
using namespace wnp;

int add_tensor(wn_plan* p, int ndim, int64_t s0, int64_t s1, int64_t s2, int is_kernel) {
  TensorInfo t;
  t.off = p->nparams;
  t.ndim = ndim;
  t.shape[0] = s0; t.shape[1] = s1; t.shape[2] = s2;
  t.len = s0 * (ndim > 1 ? s1 : 1) * (ndim > 2 ? s2 : 1);
  t.is_kernel = is_kernel;
  p->nparams += t.len;
  p->tensors.push_back(t);
  return (int)p->tensors.size() - 1;
}

ConvInfo add_conv(wn_plan* p, int taps, int cin, int cout, int dil) {
  ConvInfo c;
  c.taps = taps; c.cin = cin; c.cout = cout; c.dil = dil;
  c.kernel_t = add_tensor(p, 3, taps, cin, cout, 1);
  c.bias_t = add_tensor(p, 1, cout, 1, 1, 0);
  return c;
}

ConvInfo add_dense(wn_plan* p, int cin, int cout) {
  ConvInfo c;
  c.taps = 1; c.cin = cin; c.cout = cout; c.dil = 1;
  c.kernel_t = add_tensor(p, 2, cin, cout, 1, 1);
  c.bias_t = add_tensor(p, 1, cout, 1, 1, 0);
  return c;
}

void add_images(wn_plan* p, ConvInfo& c, bool fwd, bool bwd) {
  const TensorInfo& k = p->tensors[c.kernel_t];
  if (fwd) {
    c.fragF = p->frag_floats;
    c.fragF_stride = (int64_t)wn_frag_floats(c.cout, c.cin);
    for (int t = 0; t < c.taps; ++t) {
      WnPrepDesc d;
      memset(&d, 0, sizeof(d));
      d.src_off = k.off + (int64_t)t * c.cin * c.cout;
      d.dst_off = c.fragF + t * c.fragF_stride;
      d.I = c.cout; d.KK = c.cin; d.ld = c.cout; d.transpose = 1;
      d.q_off = 0; d.j_off = 0; d.JT = ceil32(c.cout);
      p->prep.push_back(d);
    }
    p->frag_floats += c.taps * c.fragF_stride;
  }
  if (bwd) {
    c.fragB = p->frag_floats;
    c.fragB_stride = (int64_t)wn_frag_floats(c.cin, c.cout);
    for (int t = 0; t < c.taps; ++t) {
      WnPrepDesc d;
      memset(&d, 0, sizeof(d));
      d.src_off = k.off + (int64_t)t * c.cin * c.cout;
      d.dst_off = c.fragB + t * c.fragB_stride;
      d.I = c.cin; d.KK = c.cout; d.ld = c.cout; d.transpose = 0;
      d.q_off = 0; d.j_off = 0; d.JT = ceil32(c.cin);
      p->prep.push_back(d);
    }
    p->frag_floats += c.taps * c.fragB_stride;
  }
}

// forward fp16 hi/lo split image A[cout][taps*cin] for the split-precision block kernel
void add_image16(wn_plan* p, ConvInfo& c) {
  const TensorInfo& k = p->tensors[c.kernel_t];
  c.frag16 = p->frag_floats;
  for (int t = 0; t < c.taps; ++t) {
    WnPrepDesc d;
    memset(&d, 0, sizeof(d));
    d.src_off = k.off + (int64_t)t * c.cin * c.cout;
    d.dst_off = c.frag16;
    d.I = c.cout; d.KK = c.cin; d.ld = c.cout; d.transpose = 1;
    d.q_off = t * (c.cin / 16); d.j_off = 0; d.JT = ceil32(c.cout); d.kind = 1;
    p->prep.push_back(d);
  }
  p->frag_floats += (int64_t)wn_frag16_floats(c.cout, c.taps * c.cin);
}

// generic fp16 split image made of pieces concatenated along k
int64_t new_image16(wn_plan* p, int I, int Ktotal) {
  const int64_t off = p->frag_floats;
  p->frag_floats += (int64_t)wn_frag16_floats(I, Ktotal);
  return off;
}
void add_piece16(wn_plan* p, int64_t img, int I, int64_t src_off, int KK, int ld, int transpose, int ks_off,
                 int j_off = 0, int JT_img = 0) {
  WnPrepDesc d;
  memset(&d, 0, sizeof(d));
  d.src_off = src_off; d.dst_off = img; d.I = I; d.KK = KK; d.ld = ld; d.transpose = transpose;
  d.q_off = ks_off; d.j_off = j_off; d.JT = JT_img > 0 ? JT_img : ceil32(I); d.kind = 1;
  p->prep.push_back(d);
}


int jobs_for(int K, int N) {
  return ((K + wn_wgrad_tile_k() - 1) / wn_wgrad_tile_k()) * ((N + wn_wgrad_tile_n() - 1) / wn_wgrad_tile_n());
}

int count_jobs(const wn_plan* p) {
  int n = p->KS * jobs_for(1, p->R);
  for (const BlockInfo& b : p->blocks) {
    n += p->KS * jobs_for(p->R, 2 * p->D) + jobs_for(p->D, p->R);
    for (int i = 0; i + 1 < p->LPB; ++i) n += p->KS * jobs_for(i == 0 ? p->R : p->D, p->D);
    if (b.has_skip && p->c.use_skip) n += jobs_for(p->D, p->S);
  }
  for (const ConvInfo& c : p->finals) n += jobs_for(c.cin, c.cout);
  return n;
}

}  // namespace

namespace wnp {

int64_t slab_need(int B, int T, int K, int N) {
  const int sp = wn_wgrad_choose_splits(B, T, K, N);
  return (int64_t)B * sp * ((int64_t)K * N + N);
}

// thread-local: the execution state this host thread has bound (wn_exec_bind), or null
static thread_local wn_exec* g_bound_exec = nullptr;
wn_exec& ex(const wn_plan* p) {
  return (g_bound_exec && g_bound_exec->plan == p) ? *g_bound_exec : *p->own;
}

void exec_release(wn_exec* e) {
  if (!e) return;
  for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->stack_ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->foldprep_ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->phase_ev) if (ev) (void)hipEventDestroy(ev);
  if (e->d_jobs) (void)hipFree(e->d_jobs);
  if (e->d_cov) (void)hipFree(e->d_cov);
  if (e->d_wgl) (void)hipFree(e->d_wgl);
  if (e->d_wgli) (void)hipFree(e->d_wgli);
  if (e->d_pairs) (void)hipFree(e->d_pairs);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_ffork) (void)hipEventDestroy(e->ev_ffork);
  if (e->ev_fjoin) (void)hipEventDestroy(e->ev_fjoin);
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->d_gen) (void)hipFree(e->d_gen);
  if (g_bound_exec == e) g_bound_exec = nullptr;
  delete e;
}

// device copies of the plan's immutable tables: made once, by whichever caller needs them first
int ensure_device_tables(wn_plan* p) {
  std::lock_guard<std::mutex> lock(p->tables_mu);
  if (p->tables_ready) return WN_OK;
  WN_HIP_CHECK(hipMalloc((void**)&p->d_prep, p->prep.size() * sizeof(WnPrepDesc)));
  WN_HIP_CHECK(hipMemcpy(p->d_prep, p->prep.data(), p->prep.size() * sizeof(WnPrepDesc), hipMemcpyHostToDevice));
  WN_HIP_CHECK(hipMalloc((void**)&p->d_tdesc, p->tdesc.size() * sizeof(WnTensorDesc)));
  WN_HIP_CHECK(hipMemcpy(p->d_tdesc, p->tdesc.data(), p->tdesc.size() * sizeof(WnTensorDesc), hipMemcpyHostToDevice));
  if (!p->kdesc.empty()) {
    WN_HIP_CHECK(hipMalloc((void**)&p->d_kdesc, p->kdesc.size() * sizeof(WnTensorDesc)));
    WN_HIP_CHECK(hipMemcpy(p->d_kdesc, p->kdesc.data(), p->kdesc.size() * sizeof(WnTensorDesc), hipMemcpyHostToDevice));
  }
  if (!p->prep2.empty()) {
    WN_HIP_CHECK(hipMalloc((void**)&p->d_prep2, p->prep2.size() * sizeof(WnPrepDesc)));
    WN_HIP_CHECK(hipMemcpy(p->d_prep2, p->prep2.data(), p->prep2.size() * sizeof(WnPrepDesc), hipMemcpyHostToDevice));
    WnTensorDesc d;
    d.off = 0; d.len = (int64_t)p->N * p->D * p->fold_F0 + p->fold_F0;
    WN_HIP_CHECK(hipMalloc((void**)&p->d_cov_fold, sizeof(WnTensorDesc)));
    WN_HIP_CHECK(hipMemcpy(p->d_cov_fold, &d, sizeof(WnTensorDesc), hipMemcpyHostToDevice));
    p->h_cov_fold = d;
  }
  p->tables_ready = true;
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// workspace carving

// Training passes fold the skip path into the head's first convolution when the plan has the images for it
// (fold_F0 > 0: depth-1 blocks with skip convs feeding a head whose first conv is narrower than the skip width) and the
// split-precision kernels run (the exact-fp32 mode keeps the reference's two-step form: skip sum, then the head conv).
// stacks deeper than 1 conv, TRAINING passes only: every conv of the stack on the split-precision kernels
// (inference and generation run the exact-fp32 composed kernels)
bool deep16(const wn_plan* p) { return p->deep16_ok && wn_debug_get(1) != 1; }

// The conditioning path works on [B][width] matrices (B = utterances): a rows-GEMM launch per Dense / conv is one wave
// walking its k-steps behind a global round trip each (30-250 us for a few kFLOP).  They run on the small fp32 product
// kernel instead (wn_sgemm_small32_kernel: batched over blocks, Dense epilogue) when the blocks' tensors are evenly spaced.
bool cond_small(const wn_plan* p) {
  if (p->c.cond_inputs <= 0) return false;
  for (int b = 1; b < p->N; ++b) {            // the blocks' conditioning convs must be evenly spaced in the flat buffer
    const int64_t w0 = p->tensors[p->blocks[0].conv_cond.kernel_t].off, w1 = p->tensors[p->blocks[1].conv_cond.kernel_t].off;
    if (!p->blocks[b].has_cond || p->tensors[p->blocks[b].conv_cond.kernel_t].off != w0 + (int64_t)b * (w1 - w0)) return false;
  }
  return p->blocks[0].has_cond;
}

bool fold_ok(const wn_plan* p) { return p->fold_F0 > 0 && wn_debug_get(1) != 1; }

// the head layers' weight gradients run as staged pair jobs when every final layer has a pair kind (widths 128 / 256)
// in split-precision mode
bool head_pairs_ok(const wn_plan* p) {
  if (p->finals.empty() || wn_debug_get(1) == 1) return false;
  // (folded: the first conv's gradients come from M.)  Layers without a pair kind -- e.g. the 30-column output conv of a
  // mixture head -- stay on the generic job table, on the same compact slab
  for (size_t i = fold_ok(p) ? 1 : 0; i < p->finals.size(); ++i)
    if (wn_wgrad_pair_kind(p->finals[i].cin, p->finals[i].cout) != 0) return true;
  return false;
}

WsLayout make_layout(const wn_plan* p, int B, int T, bool training) {
  WsLayout L;
  Carver cv;
  const int64_t rows = (int64_t)B * T;
  L.frag = cv.take(p->frag_floats);
  L.bias_sum = cv.take(p->Hin);
  // (fixed offsets: the prep2 table addresses V relative to the workspace base, whatever B and T are)
  L.vfold = cv.take(p->fold_F0 > 0 ? (int64_t)(p->N * p->D + 1) * p->fold_F0 : 0);     // (+ the row W_f0^T sum b_s)
  L.bfold = cv.take(p->fold_F0 > 0 ? p->fold_F0 : 0);
  L.wsall = cv.take(p->fold_F0 > 0 ? (int64_t)(p->N * p->D + 1) * p->S : 0);
  L.mslab = L.mtot = L.ytmp = 0;
  const int nH = training ? p->N + 1 : 2;
  const int hc = std::max(p->R, p->D);
  for (int i = 0; i < nH; ++i) L.H.push_back(cv.take(rows * hc));
  L.P.resize(p->N);
  if (p->LPB > 1) {
    const int nP = training ? p->N : 1;
    std::vector<std::vector<int64_t>> tmp(nP);
    for (int b = 0; b < nP; ++b)
      for (int i = 0; i < p->LPB - 1; ++i) tmp[b].push_back(cv.take(rows * p->D));
    for (int b = 0; b < p->N; ++b) L.P[b] = tmp[training ? b : 0];
  }
  L.Z = cv.take(rows * p->N * p->Dp);
  if (training) for (int b = 0; b < p->N; ++b) L.AG.push_back(cv.take(rows * p->D));
  L.U = cv.take(rows * 2 * p->D);
  L.O = cv.take(rows * p->R);
  L.skipsum = cv.take(rows * p->Hin);
  int maxC = std::max(p->Hin, p->Cout);
  for (size_t i = 0; i + 1 < p->finals.size(); ++i) {
    L.HA.push_back(cv.take(rows * p->finals[i].cout));
    maxC = std::max(maxC, p->finals[i].cout);
  }
  L.logits = cv.take(rows * p->Cout);
  L.probs = cv.take(rows * p->Cout);
  L.target = cv.take(rows);
  L.loss_rows = cv.take(rows);
  L.yt = cv.take(rows);
  L.sum_scratch = cv.take(2048 + 64);
  L.n_absmax = (int)p->finals.size() + 1 + p->N + (p->N + 1) + p->N * (p->LPB - 1);   // ... | GP[b][i] (deep stacks)
  L.absmax = cv.take(L.n_absmax);
  L.fwd_absmax = cv.take(1);
  // conditioning
  if (p->c.cond_inputs > 0) {
    for (size_t j = 0; j < p->mapping.size(); ++j) L.M.push_back(cv.take((int64_t)B * p->mapping[j].cout));
    L.cb = cv.take((int64_t)p->N * B * 2 * p->D);
    L.dcb = cv.take((int64_t)B * 2 * p->D);
    L.cbt = cv.take((int64_t)B * p->N * 2 * p->D);
    int mw = std::max(p->Cc, p->c.cond_inputs);
    for (auto& m : p->mapping) mw = std::max(mw, m.cout);
    L.g_m0 = cv.take((int64_t)B * mw);
    L.g_m1 = cv.take((int64_t)B * mw);
  } else {
    L.cb = L.dcb = L.g_m0 = L.g_m1 = L.cbt = 0;
  }
  L.gxd = 0;
  if (training && wnp::ex(p).drop_rate > 0.f) {
    for (int b = 0; b < p->N; ++b) L.XD.push_back(cv.take(rows * p->R));
    L.gxd = cv.take(rows * p->R);
  }
  if (training) {
    L.g_skipsum = cv.take(rows * p->Hin);
    int64_t need = 0;
    need = std::max(need, slab_need(B, T, 1, p->R));
    for (const BlockInfo& b : p->blocks) {
      for (const ConvInfo& c : b.dil) need = std::max(need, slab_need(B, T, c.cin, c.cout));
      need = std::max(need, slab_need(B, T, p->D, p->R));
      if (b.has_skip) need = std::max(need, slab_need(B, T, p->D, p->S));
    }
    for (const ConvInfo& c : p->finals) need = std::max(need, slab_need(B, T, c.cin, c.cout));
    if (p->fold_F0 > 0) need = std::max(need, slab_need(1, p->N * p->D + 1, p->S, p->fold_F0));
    for (const ConvInfo& c : p->mapping) need = std::max(need, slab_need(1, B, c.cin, c.cout));
    if (p->c.cond_inputs > 0) {
      need = std::max(need, slab_need(1, B, p->Cc, 2 * p->D));
      need = std::max(need, wn_colsum_scratch_floats(B, 2 * p->D));
    }
    L.slab_floats = need;
    L.slab = cv.take(need);
    L.bslab = 0; L.bsplits = 0;
    L.hslab = 0; L.hsplits = 0; L.head_base = 0; L.head_span = 0;
    L.islab = 0; L.isplits = 0;
    for (int b = 0; b < p->N; ++b) L.GU.push_back(cv.take(rows * 2 * p->D));
    // depth > 1: gradient w.r.t. the pre-activation output of every non-gated conv (operand of its weight gradient)
    L.GP.assign(p->N, std::vector<int64_t>());
    for (int b = 0; b < p->N; ++b)
      for (int i = 0; i + 1 < p->LPB; ++i) L.GP[b].push_back(cv.take(rows * p->D));
    for (int b = 0; b <= p->N; ++b) L.GH.push_back(cv.take(rows * p->R));
    if (p->S == 0) for (int b = 0; b < p->N; ++b) L.GO.push_back(cv.take(rows * p->R));
    for (size_t i = 0; i < p->finals.size(); ++i) L.GF.push_back(cv.take(rows * p->finals[i].cout));
    const int nj = count_jobs(p);
    int sp = (int)((5000 + (int64_t)nj * B - 1) / ((int64_t)nj * B));
    const int maxsp = std::max(1, (T + 255) / 256);
    sp = std::max(1, std::min(sp, maxsp));
    L.bsplits = sp;
    L.bslab = cv.take((int64_t)B * sp * p->nparams);
    if (p->fold_F0 > 0) {
      const int64_t pm = (int64_t)p->N * p->D * p->fold_F0 + p->fold_F0;
      L.mslab = cv.take((int64_t)B * sp * pm);
      L.mtot = cv.take(pm);
      L.ytmp = cv.take((int64_t)(p->N * p->D + 1) * p->S);
    }
    if (wn_inconv_wgrad_supported(p->R, p->KS) && p->tensors[p->causal.kernel_t].off == 0 &&
        p->tensors[p->causal.bias_t].off == (int64_t)p->KS * p->R) {
      L.isplits = std::max(1, std::min((512 + B - 1) / B, std::max(1, T / 64)));
      L.islab = cv.take((int64_t)B * L.isplits * (p->KS + 1) * p->R);
    }
    // The head's few products (4-8 jobs each) cannot fill the chip at the blocks' split count (one wave per
    // SIMD with one chunk of look-ahead is latency bound): ~1.5 waves per SIMD for them.  Measured at
    // configs[1] (same-process sweep, ms per step): shared 7.64 | 8 splits 7.68 | 12: 7.57 | 16: 7.71 |
    // 24: 7.63 | 32: 7.67 | 48: 7.71 -- a shallow, irregular optimum.
    if (!p->finals.empty()) {
      int njh = 0;
      for (const ConvInfo& c : p->finals) njh += jobs_for(c.cin, c.cout);
      int hs = (int)((1536 + (int64_t)njh * B - 1) / ((int64_t)njh * B));
      // pair jobs: one workgroup per (layer, utterance, time range) -> about one workgroup per CU and layer
      if (head_pairs_ok(p)) hs = std::max(1, (256 + B - 1) / B);
      hs = std::max(sp, std::min(hs, maxsp));
      if (hs > sp) {
        L.head_base = p->tensors[p->finals.front().kernel_t].off;
        const TensorInfo& last = p->tensors[p->finals.back().bias_t];
        L.head_span = last.off + last.len - L.head_base;
        L.hsplits = hs;
        L.hslab = cv.take((int64_t)B * hs * L.head_span);
      }
    }
  } else {
    L.bslab = 0; L.bsplits = 0;
    L.hslab = 0; L.hsplits = 0; L.head_base = 0; L.head_span = 0;
    L.islab = 0; L.isplits = 0;
    L.g_skipsum = L.slab = 0;
    L.slab_floats = 0;
  }
  L.total = cv.pos;
  return L;
}

}  // namespace wnp

using namespace wnp;

// ==========================================================================================
// C-ABI: plan
// ==========================================================================================
extern "C" wn_plan* wn_plan_create(const wn_config* cfg) {
  if (!cfg) { wn_set_error("plan_create: null config"); return nullptr; }
  const wn_config& c = *cfg;
  // src/model.py:52-70 argument validation (the Python layer raises ValueError earlier)
  if (c.kernel_size < 2) { wn_set_error("Kernel size must be at least 2."); return nullptr; }
  if (c.layers_per_block < 1 || c.layers_per_block > 16) { wn_set_error("Layers per block must be in 1..16."); return nullptr; }
  if (c.blocks < 1) { wn_set_error("Blocks must be at least 1."); return nullptr; }
  if (c.channels < 1) { wn_set_error("channels must be positive"); return nullptr; }
  {
    const double lg = log((double)c.dilation_bound) / log((double)c.kernel_size);
    int64_t pw = 1; int mp = 0;
    while (pw < c.dilation_bound) { pw *= c.kernel_size; ++mp; }
    if (pw != c.dilation_bound || mp < 1) { wn_set_error("dilation bound must be power of kernel_size."); return nullptr; }
    (void)lg;
  }
  if (c.head == WN_HEAD_CATEGORICAL && c.num_mixtures != 0) { wn_set_error("Categorical sampling cannot be used with mixtures."); return nullptr; }
  if (c.head != WN_HEAD_CATEGORICAL && c.num_mixtures < 1) { wn_set_error("Number of mixtures must be at least 1 for mixture heads."); return nullptr; }
  if (c.n_final < 0 || c.n_final > WN_MAX_FINAL || c.n_mapping < 0 || c.n_mapping > WN_MAX_MAPPING) { wn_set_error("too many final / mapping layers"); return nullptr; }
  if (c.bits < 1 || c.bits > 16) { wn_set_error("bits must be in 1..16"); return nullptr; }
  if (c.kernel_size > 3) { wn_set_error("kernel_size > 3 is not supported by the gfx950 kernels"); return nullptr; }

  wn_plan* p = new wn_plan();
  p->c = c;
  p->KS = c.kernel_size; p->R = c.channels; p->D = c.dilation_channels > 0 ? c.dilation_channels : c.channels;
  p->S = c.skip_channels; p->N = c.blocks; p->LPB = c.layers_per_block;
  p->Cout = c.num_mixtures > 0 ? 3 * c.num_mixtures : (1 << c.bits);
  p->Sh = p->S > 0 ? p->S : p->R;
  p->Hin = c.use_skip ? p->Sh : p->R;
  p->Dp = ceil8(p->D);
  // dilation schedule, src/model.py:79-81
  {
    int mp = 0; int64_t pw = 1;
    while (pw < c.dilation_bound) { pw *= c.kernel_size; ++mp; }
    for (int i = 0; i < p->LPB * p->N; ++i) {
      int d = 1;
      for (int e = 0; e < i % mp; ++e) d *= c.kernel_size;
      p->dilations.push_back(d);
    }
  }
  // mapped condition width, src/model.py:141-148
  p->Cc = 0;
  if (c.cond_inputs > 0) p->Cc = c.n_mapping > 0 ? c.mapping_channels[c.n_mapping - 1] : c.cond_inputs;

  // ---- tensors in Keras creation order ----
  p->causal = add_conv(p, p->KS, 1, p->R, 1);
  for (int b = 0; b < p->N; ++b) {
    BlockInfo bi;
    int cin = p->R;
    for (int i = 0; i < p->LPB; ++i) {
      const int cout = (i == p->LPB - 1) ? 2 * p->D : p->D;
      bi.dil.push_back(add_conv(p, p->KS, cin, cout, p->dilations[b * p->LPB + i]));
      cin = cout;
    }
    bi.conv1 = add_conv(p, 1, p->D, p->R, 1);
    bi.has_skip = p->S > 0;
    if (bi.has_skip) bi.conv_skip = add_conv(p, 1, p->D, p->S, 1);
    bi.has_cond = c.cond_inputs > 0;
    if (bi.has_cond) bi.conv_cond = add_conv(p, 1, p->Cc, 2 * p->D, 1);
    p->blocks.push_back(bi);
  }
  {
    int cprev = p->Hin;
    for (int i = 0; i < c.n_final; ++i) { p->finals.push_back(add_conv(p, 1, cprev, c.final_channels[i], 1)); cprev = c.final_channels[i]; }
    p->finals.push_back(add_conv(p, 1, cprev, p->Cout, 1));
  }
  if (c.cond_inputs > 0) {
    int cin = c.cond_inputs;
    for (int j = 0; j < c.n_mapping; ++j) { p->mapping.push_back(add_dense(p, cin, c.mapping_channels[j])); cin = c.mapping_channels[j]; }
  }
  for (const TensorInfo& t : p->tensors) {
    WnTensorDesc d; d.off = t.off; d.len = t.len;
    p->tdesc.push_back(d);
    if (t.is_kernel) p->kdesc.push_back(d);
  }

  // ---- weight images ----
  add_images(p, p->causal, true, false);
  for (BlockInfo& bi : p->blocks) {
    for (ConvInfo& cv : bi.dil) add_images(p, cv, true, true);
    add_images(p, bi.conv1, true, true);
    if (bi.has_skip) add_images(p, bi.conv_skip, false, true);
    if (bi.has_cond) add_images(p, bi.conv_cond, true, true);
  }
  for (ConvInfo& cv : p->finals) add_images(p, cv, true, true);
  for (ConvInfo& cv : p->mapping) add_images(p, cv, true, true);
  {
    bool all_cond = p->c.cond_inputs > 0 && (2 * p->D) % 32 == 0 && p->N > 0;
    for (const BlockInfo& bi : p->blocks) all_cond = all_cond && bi.has_cond;
    if (all_cond) {
      const int D2 = 2 * p->D;
      p->frag_condF = p->frag_floats;
      for (int b = 0; b < p->N; ++b) {
        WnPrepDesc d;
        memset(&d, 0, sizeof(d));
        d.src_off = p->tensors[p->blocks[b].conv_cond.kernel_t].off;
        d.dst_off = p->frag_condF;
        d.I = D2; d.KK = p->Cc; d.ld = D2; d.transpose = 1;
        d.q_off = 0; d.j_off = b * (D2 / 32); d.JT = p->N * (D2 / 32);
        p->prep.push_back(d);
      }
      p->frag_floats += (int64_t)wn_frag_floats(p->N * D2, p->Cc);
      p->frag_condB = p->frag_floats;
      for (int b = 0; b < p->N; ++b) {
        WnPrepDesc d;
        memset(&d, 0, sizeof(d));
        d.src_off = p->tensors[p->blocks[b].conv_cond.kernel_t].off;
        d.dst_off = p->frag_condB;
        d.I = p->Cc; d.KK = D2; d.ld = D2; d.transpose = 0;
        d.q_off = b * (D2 / 8); d.j_off = 0; d.JT = ceil32(p->Cc);
        p->prep.push_back(d);
      }
      p->frag_floats += (int64_t)wn_frag_floats(p->Cc, p->N * D2);
    }
  }
  // folded skip sum: A[Sh][N*Dp], piece b = (conv_skip or conv1 of block b)^T
  p->frag_skipF = p->frag_floats;
  for (int b = 0; b < p->N; ++b) {
    const ConvInfo& src = p->blocks[b].has_skip ? p->blocks[b].conv_skip : p->blocks[b].conv1;
    WnPrepDesc d;
    memset(&d, 0, sizeof(d));
    d.src_off = p->tensors[src.kernel_t].off;
    d.dst_off = p->frag_skipF;
    d.I = p->Sh; d.KK = p->D; d.ld = p->Sh; d.transpose = 1;
    d.q_off = b * (p->Dp / 8); d.j_off = 0; d.JT = ceil32(p->Sh);
    p->prep.push_back(d);
  }
  p->frag_floats += (int64_t)wn_frag_floats(p->Sh, p->N * p->Dp);
  p->fused_ok = wn_layer_fwd_supported(p->R, p->D, p->KS) != 0;
  p->fused16_ok = p->fused_ok && wn_layer_fwd_f16_supported(p->R, p->D, p->KS) != 0;
  if (p->fused16_ok)
    for (BlockInfo& bi : p->blocks) {
      add_image16(p, bi.dil.back());
      add_image16(p, bi.conv1);
    }
  // blocks too wide for the one-kernel forward (R = D = 128): gated conv with the gate in the GEMM epilogue,
  // then the 1x1 convolution with the residual add -- both on the split-precision streamed GEMM
  if (!p->fused16_ok && p->LPB == 1 && p->D % 64 == 0 && m16(p->R) && m32(p->R))
    for (BlockInfo& bi : p->blocks) {
      const ConvInfo& c = bi.dil.back();
      bi.f16gate = new_image16(p, 2 * p->D, p->KS * p->R);
      for (int t = 0; t < p->KS; ++t)
        for (int q = 0; q < p->D / 64; ++q)
          for (int half = 0; half < 2; ++half)
            add_piece16(p, bi.f16gate, 64, p->tensors[c.kernel_t].off + (int64_t)t * p->R * 2 * p->D + half * p->D + 64 * q,
                        p->R, 2 * p->D, 1, t * (p->R / 16), 4 * q + 2 * half, 2 * p->D / 32);
      add_image16(p, bi.conv1);
      // the streamed-weights one-kernel forward (wn_layer16s.hip) reads the conv's image in natural row-tile order
      if (wn_layer_fwd_s128_supported(p->R, p->D, p->KS)) {
        bi.f16nat = new_image16(p, 2 * p->D, p->KS * p->R);
        for (int t = 0; t < p->KS; ++t)
          add_piece16(p, bi.f16nat, 2 * p->D, p->tensors[c.kernel_t].off + (int64_t)t * p->R * 2 * p->D, p->R, 2 * p->D, 1,
                      t * (p->R / 16));
      }
    }
  // split-precision images of the generic contractions (each only when its shape qualifies:
  // K multiple of 16, N multiple of 32 and >= 64; otherwise the fp32-MFMA kernel runs)
  for (ConvInfo& c : p->finals) {
    if (m16(c.cin) && m32(c.cout)) add_image16(p, c);
    if (m16(c.cout) && m32(c.cin)) {
      c.frag16B = new_image16(p, c.cin, c.cout);
      add_piece16(p, c.frag16B, c.cin, p->tensors[c.kernel_t].off, c.cout, c.cout, 0, 0);
    }
  }
  if (p->c.use_skip && m16(p->D) && p->Dp == p->D && m32(p->Sh)) {
    p->frag16_skipF = new_image16(p, p->Sh, p->N * p->D);
    for (int b = 0; b < p->N; ++b) {
      const ConvInfo& src = p->blocks[b].has_skip ? p->blocks[b].conv_skip : p->blocks[b].conv1;
      add_piece16(p, p->frag16_skipF, p->Sh, p->tensors[src.kernel_t].off, p->D, p->Sh, 1, b * (p->D / 16));
    }
  }
  // Stacks deeper than 1 conv (layers_per_block > 1; the reference's default network is 5 x 5, train.py:31-49): training
  // passes run every conv of the stack, forward and backward, on the split-precision kernels.  Widths of 32 are padded
  // to two row tiles.  (Inference and generation keep the exact-fp32 composed kernels: the queued sampler reproduces them
  // bit for bit.)
  if (p->LPB > 1 && p->fused16_ok && p->R == p->D && p->R % 32 == 0 && p->Dp == p->D && (p->S == 0 || m16(p->S))) {
    auto padded = [&](int I, int K) { return new_image16(p, std::max(I, 64), K); };
    auto jt_of = [&](int I) { return std::max(2, ceil32(I)); };
    for (BlockInfo& bi : p->blocks) {
      for (int i = 0; i < p->LPB; ++i) {
        const ConvInfo& c = bi.dil[i];
        if (i + 1 < p->LPB) {                      // forward: A[n][tap * cin + k] = W[tap][k][n]
          bi.d16F[i] = padded(c.cout, p->KS * c.cin);
          for (int t = 0; t < p->KS; ++t)
            add_piece16(p, bi.d16F[i], c.cout, p->tensors[c.kernel_t].off + (int64_t)t * c.cin * c.cout, c.cin, c.cout, 1,
                        t * (c.cin / 16), 0, jt_of(c.cout));
        }
        // backward data: A[k][tap * cout + n] = W[tap][k][n]
        bi.d16B[i] = padded(c.cin, p->KS * c.cout);
        for (int t = 0; t < p->KS; ++t)
          add_piece16(p, bi.d16B[i], c.cin, p->tensors[c.kernel_t].off + (int64_t)t * c.cin * c.cout, c.cout, c.cout, 0,
                      t * (c.cout / 16), 0, jt_of(c.cin));
      }
      // d z = W_r g_o + W_s g_skip : image [W_r | W_s], I = D
      bi.g16u = padded(p->D, p->R + p->S);
      add_piece16(p, bi.g16u, p->D, p->tensors[bi.conv1.kernel_t].off, p->R, p->R, 0, 0, 0, jt_of(p->D));
      if (bi.has_skip) add_piece16(p, bi.g16u, p->D, p->tensors[bi.conv_skip.kernel_t].off, p->S, p->S, 0, p->R / 16, 0, jt_of(p->D));
    }
    p->deep16_ok = true;
  }
  if (p->LPB == 1 && m32(p->D) && m16(p->R) && (p->S == 0 || m16(p->S))) {
    for (BlockInfo& bi : p->blocks) {
      // d z = W_r g_o + W_s g_skip : image [W_r | W_s], I = D
      bi.g16u = new_image16(p, p->D, p->R + p->S);
      add_piece16(p, bi.g16u, p->D, p->tensors[bi.conv1.kernel_t].off, p->R, p->R, 0, 0);
      if (bi.has_skip) add_piece16(p, bi.g16u, p->D, p->tensors[bi.conv_skip.kernel_t].off, p->S, p->S, 0, p->R / 16);
    }
  }
  // skip path folded into the head's first conv: needs skip convs feeding a head with >= 1 hidden layer narrower than S
  if (p->LPB == 1 && p->c.use_skip && p->S > 0 && p->finals.size() >= 2 && p->finals[0].cout < p->S && m32(p->finals[0].cout) &&
      m32(p->D) && p->D % 64 == 0 && m16(p->R) && m16(p->S) && p->Dp == p->D && p->frag16_skipF >= 0 &&
      wn_wgrad_skip_supported(p->D, p->finals[0].cout, p->N * p->D) && p->finals[0].frag16B >= 0) {
    const int F0 = p->finals[0].cout;
    p->fold_F0 = F0;
    // offsets of V inside the workspace (make_layout: images, bias_sum, V, ...) -- fixed for the plan
    // (frag_floats is still growing here: the prep2 descriptors are finished at the end of this function)
    p->frag16_foldF = new_image16(p, F0, p->N * p->D);
    for (BlockInfo& bi : p->blocks) {
      bi.g16uf = new_image16(p, p->D, p->R + F0);
      add_piece16(p, bi.g16uf, p->D, p->tensors[bi.conv1.kernel_t].off, p->R, p->R, 0, 0);
    }
  }
  if (p->LPB == 1 && m32(p->R) && m16(2 * p->D)) {
    for (BlockInfo& bi : p->blocks) {
      // d x = sum_tap W_tap g_u[t + shift] : image of KS pieces A[R][2D], I = R
      ConvInfo& c = bi.dil[0];
      c.frag16B = new_image16(p, p->R, p->KS * 2 * p->D);
      for (int t = 0; t < p->KS; ++t)
        add_piece16(p, c.frag16B, p->R, p->tensors[c.kernel_t].off + (int64_t)t * p->R * 2 * p->D, 2 * p->D,
                    2 * p->D, 0, t * (2 * p->D / 16));
    }
  }
  if (p->fold_F0 > 0) {
    // pieces read from the workspace: V = [N*D][F0] at the fixed offset make_layout gives it
    const int F0 = p->fold_F0;
    const int64_t voff = make_layout(p, 1, 1, false).vfold;
    auto piece2 = [&](int64_t img, int I, int64_t src_off, int KK, int ld, int transpose, int ks_off) {
      WnPrepDesc d;
      memset(&d, 0, sizeof(d));
      d.src_off = src_off; d.dst_off = img; d.I = I; d.KK = KK; d.ld = ld; d.transpose = transpose;
      d.q_off = ks_off; d.j_off = 0; d.JT = ceil32(I); d.kind = 1;
      p->prep2.push_back(d);
    };
    piece2(p->frag16_foldF, F0, voff, p->N * p->D, F0, 1, 0);                    // A[n][k] = V[k][n]
    for (int b = 0; b < p->N; ++b)                                               // A[c][R + n] = V[b*D + c][n]
      piece2(p->blocks[b].g16uf, p->D, voff + (int64_t)b * p->D * F0, F0, F0, 0, p->R / 16);
  }
  p->own = new wn_exec;
  p->own->plan = p;
  return p;
}

extern "C" void wn_plan_destroy(wn_plan* p) {
  if (!p) return;
  if (p->d_prep) (void)hipFree(p->d_prep);
  if (p->d_tdesc) (void)hipFree(p->d_tdesc);
  if (p->d_kdesc) (void)hipFree(p->d_kdesc);
  if (p->d_prep2) (void)hipFree(p->d_prep2);
  if (p->d_cov_fold) (void)hipFree(p->d_cov_fold);
  wnp::exec_release(p->own);
  delete p;
}

// ---- execution states: one per host thread / stream that drives the same plan (see struct wn_exec) ----
extern "C" wn_exec* wn_exec_create(const wn_plan* p) {
  if (!p) { wn_set_error("exec_create: null plan"); return nullptr; }
  wn_exec* e = new wn_exec;
  e->plan = p;
  // a new state starts from the plan's own dropout setting and phase selection (what the Python layer configured)
  e->drop_rate = p->own->drop_rate; e->drop_seed = p->own->drop_seed; e->drop_step = p->own->drop_step;
  e->train_phases = p->own->train_phases;
  return e;
}
extern "C" void wn_exec_destroy(wn_exec* e) { wnp::exec_release(e); }
extern "C" int wn_exec_bind(wn_exec* e) {
  wnp::g_bound_exec = e;
  return WN_OK;
}

// ---- Dropout(rate) on every block input in training mode (src/layers.py:108-111, 195-196) ----
extern "C" int wn_plan_set_dropout(wn_plan* p, float rate, uint64_t seed, uint64_t step) {
  if (!p || rate < 0.f || rate >= 1.f) { wn_set_error("Dropout must be between 0 and 1."); return WN_E_INVALID; }
  wnp::ex(p).drop_rate = rate; wnp::ex(p).drop_seed = seed; wnp::ex(p).drop_step = step;
  return WN_OK;
}
extern "C" uint32_t wn_dropout_key_for(uint64_t seed, int32_t block, uint64_t step) { return wn_dropout_key(seed, block, step); }

// ---- profiling hook: HIP events around the residual-block forward launches.  When a forward pass runs
//      its N blocks as N back-to-back launches of the fused block kernel (nothing else in between) ONE
//      event pair brackets the chain and counts N launches: a pair per launch adds its own event packets
//      (+5 us per launch on MI355X) to what it measures.  Otherwise one pair per block. ----
extern "C" int wn_prof_enable(wn_plan* p, int32_t max_launches) {
  if (!p) return WN_E_INVALID;
  for (hipEvent_t e : wnp::ex(p).prof_ev) (void)hipEventDestroy(e);
  wnp::ex(p).prof_ev.clear();
  wnp::ex(p).prof_cnt.assign(max_launches > 0 ? max_launches : 0, 1);
  wnp::ex(p).prof_used = 0;
  wnp::ex(p).prof_on = max_launches > 0;
  for (int i = 0; i < 2 * max_launches; ++i) {
    hipEvent_t e;
    WN_HIP_CHECK(hipEventCreate(&e));
    wnp::ex(p).prof_ev.push_back(e);
  }
  return WN_OK;
}
// ---- the whole block stack: one event pair per forward pass, first block launch -> end of the folded skip sum ----
extern "C" int wn_stack_prof_enable(wn_plan* p, int32_t max_passes) {
  if (!p) return WN_E_INVALID;
  for (hipEvent_t e : wnp::ex(p).stack_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : wnp::ex(p).foldprep_ev) (void)hipEventDestroy(e);
  wnp::ex(p).stack_ev.clear();
  wnp::ex(p).foldprep_ev.clear();
  wnp::ex(p).stack_used = 0;
  wnp::ex(p).foldprep_used = 0;
  for (int i = 0; i < 2 * max_passes; ++i) {
    hipEvent_t e;
    WN_HIP_CHECK(hipEventCreate(&e));
    wnp::ex(p).stack_ev.push_back(e);
    WN_HIP_CHECK(hipEventCreate(&e));
    wnp::ex(p).foldprep_ev.push_back(e);
  }
  return WN_OK;
}
// the same passes' weight-space preparation of the folded skip path (bias sum, V = W_s W_f0, its fp16 images), which
// runs before the first block launch: average per pass, 0 when the plan does not fold.  Call BEFORE wn_stack_prof_read.
extern "C" int wn_stack_prof_read_foldprep(wn_plan* p, int32_t* passes, float* avg_ms) {
  if (!p || !passes || !avg_ms) return WN_E_INVALID;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i + 1 < wnp::ex(p).foldprep_used; i += 2) {
    float ms = 0.f;
    WN_HIP_CHECK(hipEventElapsedTime(&ms, wnp::ex(p).foldprep_ev[i], wnp::ex(p).foldprep_ev[i + 1]));
    tot += ms; ++n;
  }
  *passes = n;
  *avg_ms = n ? (float)(tot / n) : 0.f;
  wnp::ex(p).foldprep_used = 0;
  return WN_OK;
}
extern "C" int wn_stack_prof_read(wn_plan* p, int32_t* passes, float* avg_ms) {
  if (!p || !passes || !avg_ms) return WN_E_INVALID;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i + 1 < wnp::ex(p).stack_used; i += 2) {
    float ms = 0.f;
    WN_HIP_CHECK(hipEventElapsedTime(&ms, wnp::ex(p).stack_ev[i], wnp::ex(p).stack_ev[i + 1]));
    tot += ms; ++n;
  }
  *passes = n;
  *avg_ms = n ? (float)(tot / n) : 0.f;
  wnp::ex(p).stack_used = 0;
  return WN_OK;
}
// ---- phase marks of wn_train_fwd_bwd (bench.py): events after the forward, the loss, the backward-data
//      chain and at the end; wn_phase_read returns the four durations of the LAST call in milliseconds ----
extern "C" int wn_phase_enable(wn_plan* p, int32_t on) {
  if (!p) return WN_E_INVALID;
  if (on && !wnp::ex(p).phase_ev[0])
    for (hipEvent_t& e : wnp::ex(p).phase_ev) WN_HIP_CHECK(hipEventCreate(&e));
  wnp::ex(p).phase_on = on != 0;
  return WN_OK;
}
extern "C" int wn_phase_read(wn_plan* p, float* ms4) {
  if (!p || !ms4 || !wnp::ex(p).phase_ev[0]) return WN_E_INVALID;
  for (int i = 0; i < 4; ++i) WN_HIP_CHECK(hipEventElapsedTime(ms4 + i, wnp::ex(p).phase_ev[i], wnp::ex(p).phase_ev[i + 1]));
  return WN_OK;
}
// average milliseconds per recorded launch (call after the stream has been synchronised)
extern "C" int wn_prof_read(wn_plan* p, int32_t* launches, float* avg_ms) {
  if (!p || !launches || !avg_ms) return WN_E_INVALID;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i + 1 < wnp::ex(p).prof_used; i += 2) {
    float ms = 0.f;
    WN_HIP_CHECK(hipEventElapsedTime(&ms, wnp::ex(p).prof_ev[i], wnp::ex(p).prof_ev[i + 1]));
    tot += ms; n += wnp::ex(p).prof_cnt[i / 2];
  }
  *launches = n;
  *avg_ms = n ? (float)(tot / n) : 0.f;
  wnp::ex(p).prof_used = 0;
  return WN_OK;
}

extern "C" int64_t wn_plan_param_count(const wn_plan* p) { return p ? p->nparams : 0; }
extern "C" int32_t wn_plan_num_tensors(const wn_plan* p) { return p ? (int32_t)p->tensors.size() : 0; }
extern "C" int wn_plan_tensor_info(const wn_plan* p, int32_t idx, int64_t* offset, int64_t* len,
                                   int32_t* ndim, int64_t* shape3, int32_t* is_kernel) {
  if (!p || idx < 0 || idx >= (int32_t)p->tensors.size()) { wn_set_error("tensor_info: bad index"); return WN_E_INVALID; }
  const TensorInfo& t = p->tensors[idx];
  if (offset) *offset = t.off;
  if (len) *len = t.len;
  if (ndim) *ndim = t.ndim;
  if (shape3) { shape3[0] = t.shape[0]; shape3[1] = t.shape[1]; shape3[2] = t.shape[2]; }
  if (is_kernel) *is_kernel = t.is_kernel;
  return WN_OK;
}
extern "C" int32_t wn_plan_receptive_field(const wn_plan* p) {
  int64_t s = 0;
  for (int d : p->dilations) s += d;
  return (int32_t)(1 + s * (p->KS - 1) + 1);            // src/model.py:122
}
extern "C" int32_t wn_plan_out_channels(const wn_plan* p) { return p->Cout; }
extern "C" int32_t wn_plan_dilation(const wn_plan* p, int32_t i) {
  return (i >= 0 && i < (int32_t)p->dilations.size()) ? p->dilations[i] : -1;
}
extern "C" int64_t wn_plan_workspace_floats(const wn_plan* p, int32_t B, int32_t T, int32_t training) {
  if (!p || B < 1 || T < 1) return 0;
  return make_layout(p, B, T, training != 0).total;
}

// test / diagnosis hook: where a training call keeps an intermediate in the caller's workspace (float offset and
// length), so that parity tests can compare saved activations and data gradients with the oracle's.
//   what: 0 block input H[idx] (idx 0..N) | 1 gated activations Z of block idx | 2 saved sigmoid of block idx |
//         3 skip sum | 4 head activation idx | 5 logits | 6 d loss / d (final layer idx output, pre-activation) |
//         7 d loss / d skip sum | 8 d loss / d u of block idx ([rows][2D]) | 9 d loss / d H[idx] | 10 running max-abs slots
// Which kernel family each phase of a pass selects for this plan under the calling thread's switches -- the fast paths are
// shape-specialised, everything else takes composed paths that are several times slower; this makes
// the choice visible (WaveNet.kernel_report(), bench.py "kernel_families", WN_LOG_KERNELS=1 prints it once per plan).
extern "C" int wn_plan_describe(const wn_plan* p, char* buf, int32_t len) {
  if (!p || !buf || len < 1) return WN_E_INVALID;
  const bool exact = wn_debug_get(1) == 1;
  const char* fwd;
  if (p->LPB > 1 && deep16(p)) fwd = "per conv of the stack, split precision in training passes (rows contractions + the fused block kernel for the gated conv)";
  else if (p->LPB > 1) fwd = "composed per conv (rows GEMM fp32 + fused fp32 kernel for the gated conv where the shape allows)";
  else if (!exact && p->fused16_ok) fwd = "fused split-precision block kernel, weights LDS-resident (wn_layer_fwd_f16_kernel)";
  else if (!exact && !p->blocks.empty() && p->blocks[0].f16nat >= 0)
    fwd = "fused split-precision block kernel, weights streamed through an LDS ring (wn_layer_fwd_s128_kernel)";
  else if (!exact && !p->blocks.empty() && p->blocks[0].f16gate >= 0)
    fwd = "two split-precision contractions per block (gated conv + gate, 1x1 + residual)";
  else if (p->fused_ok) fwd = "fused exact-fp32 block kernel (wn_layer_fwd_kernel)";
  else fwd = "composed: rows GEMM -> gate kernel -> rows GEMM";
  const bool fold = fold_ok(p);
  const char* bwd;
  if (p->LPB > 1) bwd = "per-block composed backward, one rows contraction per conv of the stack (layers_per_block > 1)";
  else if (fold && p->N >= 2 && wn_bwd_pair_supported(p->R, p->D, p->KS, p->fold_F0) && p->Dp == p->D)
    bwd = "two products per launch (wn_bwd_pair_kernel: g_x(b+1) and g_u(b))";
  else if (fold && p->N >= 2 && wn_bwd_s128_supported(p->R, p->D, p->KS, p->fold_F0) && p->Dp == p->D)
    bwd = "two products per launch, weights streamed through an LDS ring (wn_bwd_s128_kernel: g_x(b+1) and g_u(b))";
  else if (!exact) bwd = "two split-precision rows contractions per block (g_u with the gate derivative, g_x)";
  else bwd = "two exact-fp32 rows contractions per block";
  const char* wg;
  if (p->LPB > 1 && deep16(p) && wn_wgrad_layer_supported(p->R, p->D, p->KS) && p->Dp == p->R)
    wg = "one workgroup per (conv, utterance, time range): the last conv + 1x1 of a block as for depth 1, inner convs through the kernel's INNER form (wn_wgrad_layer_kernel)";
  else if (p->LPB > 1 && deep16(p)) wg = "generic batched job table, split precision, every conv of every stack in one launch (wn_wgrad_batched_kernel)";
  else if (p->LPB > 1) wg = "generic batched job table in exact fp32, every conv of every stack in one launch (wn_wgrad_batched_kernel)";
  else if (!exact && wn_wgrad_layer_supported(p->R, p->D, p->KS) && p->Dp == p->R)
    wg = "one workgroup per (block, utterance, time range) for dW_d, db_d, dW_r, db_r (wn_wgrad_layer_kernel)";
  else if (!exact && p->KS == 2 && p->R == p->D && p->Dp == p->D &&
           wn_wgrad_pair_kind(p->R, 2 * p->D) == 1 && wn_wgrad_pair_kind(p->D, p->R) == 2)
    wg = (fold && p->fold_F0 == 128 && p->D == 128)
             ? "two jobs per block on transposed LDS reads: both taps of dW_d from one read of du; dW_r together with the "
               "folded skip path's M = Z^T dL/da from one read of z (wn_wgrad_tr_kernel)"
             : "two jobs per block on transposed LDS reads: both taps of dW_d from one read of du; dW_r (wn_wgrad_tr_kernel)";
  else wg = "generic batched job table (wn_wgrad_batched_kernel)";
  snprintf(buf, (size_t)len,
           "math: %s | block forward: %s | skip path: %s | backward data: %s | block weight gradients: %s",
           exact ? "exact fp32 MFMA" : "fp16 hi|lo split, 3 products, fp32 accumulate", fwd,
           fold ? "folded into the head's first conv (V = W_s W_f0, training / inference / generation)"
                : (p->c.use_skip ? "one contraction over all blocks' gated activations, then the head" : "none (use_skip False)"),
           bwd, wg);
  return WN_OK;
}

extern "C" int wn_debug_ws_region(const wn_plan* p, int32_t B, int32_t T, int32_t what, int32_t idx, int64_t* off,
                                  int64_t* len) {
  if (!p || !off || !len || B < 1 || T < 1) return WN_E_INVALID;
  const WsLayout L = make_layout(p, B, T, true);
  const int64_t rows = (int64_t)B * T;
  auto in = [&](size_t n) { return idx >= 0 && (size_t)idx < n; };
  *off = -1; *len = 0;
  switch (what) {
    case 0: if (in(L.H.size())) { *off = L.H[idx]; *len = rows * p->R; } break;
    case 1: if (in((size_t)p->N)) { *off = L.Z + (int64_t)idx * rows * p->Dp; *len = rows * p->Dp; } break;
    case 2: if (in(L.AG.size())) { *off = L.AG[idx]; *len = rows * p->D; } break;
    case 3: if (!fold_ok(p)) { *off = L.skipsum; *len = rows * p->Hin; } break;      // (folded training pass: never formed)
    case 4: if (in(L.HA.size())) { *off = L.HA[idx]; *len = rows * p->finals[idx].cout; } break;
    case 5: *off = L.logits; *len = rows * p->Cout; break;
    case 6: if (in(L.GF.size())) { *off = L.GF[idx]; *len = rows * p->finals[idx].cout; } break;
    case 7: if (!fold_ok(p)) { *off = L.g_skipsum; *len = rows * p->Hin; } break;
    case 8: if (in(L.GU.size())) { *off = L.GU[idx]; *len = rows * 2 * p->D; } break;
    case 9: if (in(L.GH.size())) { *off = L.GH[idx]; *len = rows * p->R; } break;
    case 10: *off = L.absmax; *len = L.n_absmax; break;
    case 11: {                           // activated output of non-gated conv i of block b (idx = b * (LPB - 1) + i), depth > 1
      const int inner = p->LPB - 1;
      if (inner > 0 && idx >= 0 && idx < p->N * inner && (size_t)(idx / inner) < L.P.size() &&
          (size_t)(idx % inner) < L.P[idx / inner].size()) { *off = L.P[idx / inner][idx % inner]; *len = rows * p->D; }
      break;
    }
    default: break;
  }
  if (*off < 0) { wn_set_error("ws_region: no such region (%d, %d)", what, idx); return WN_E_INVALID; }
  return WN_OK;
}

// forward range guard of an inference call: reads nothing back, only tells where the slot is
extern "C" int64_t wn_plan_range_slot(const wn_plan* p, int32_t B, int32_t T, int32_t training) {
  if (!p || B < 1 || T < 1) return -1;
  return make_layout(p, B, T, training != 0).fwd_absmax;
}
extern "C" float wn_range_limit(void) { return WN_RANGE_LIMIT; }
