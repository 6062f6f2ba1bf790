// ==========================================================================================
// generation: WaveNet.generate / _generation, src/model.py:241-307 (intended semantics:
// SURVEY.md section 9 item 8 -- the reference's bad kwarg / rank bugs are not reproduced)
// ==========================================================================================
#include "wn_plan_internal.h"

using namespace wnp;

namespace {

__global__ void wn_gen_shift_kernel(const float* win, const float* sample, int B, int RF, float* win_next,
                                    float* out, int length, int step) {
  const int64_t n = (int64_t)B * RF;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / RF), t = (int)(i % RF);
    win_next[i] = (t == RF - 1) ? sample[b] : win[i + 1];
    if (t == RF - 1) out[(int64_t)b * length + step] = sample[b];
  }
}

__global__ void wn_gather_last_kernel(const float* logits, int B, int RF, int C, float* last) {
  const int64_t n = (int64_t)B * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / C), c = (int)(i % C);
    last[i] = logits[((int64_t)b * RF + (RF - 1)) * C + c];
  }
}

}  // namespace

namespace {

struct GenLayout {
  int64_t prime;                       // priming forward workspace (make_layout(B, RF, inference))
  int64_t win0, win1, last, lastp, samp;
  int64_t guard;                       // range guard of the call: running max-abs of every input of a split-precision kernel
                                       // (+ 1: the relay's give-up word, see wn_gen_relay128_kernel)
  int64_t relay;                       // granule areas of the 128-channel relay, or < 0
  int64_t xin;                         // [KS][B]
  std::vector<int64_t> ring;           // per block [nslots][B][R]: inputs of the first dilated conv
  std::vector<int> nslots;
  std::vector<std::vector<int64_t>> ringp;   // layers_per_block > 1: inputs of dilated conv i + 1, [nslots][B][D]
  std::vector<std::vector<int>> nslots_p;
  int64_t Zrow, skiprow, hrow0, hrow1, dummy;   // per-step rows
  int64_t u0;                          // fused step: partial accumulators of all blocks
  std::vector<int64_t> HArow;
  int64_t total;
};

GenLayout gen_layout(const wn_plan* p, int B, bool queued) {
  GenLayout G;
  Carver cv;
  const int RF = wn_plan_receptive_field(p);
  G.prime = cv.take(make_layout(p, B, RF, false).total);
  G.win0 = cv.take((int64_t)B * RF);
  G.win1 = cv.take((int64_t)B * RF);
  G.last = cv.take((int64_t)B * p->Cout);
  G.lastp = cv.take((int64_t)B * p->Cout);
  G.samp = cv.take(B);
  G.guard = cv.take(2);
  G.relay = -1;
  G.xin = G.Zrow = G.skiprow = G.hrow0 = G.hrow1 = G.dummy = G.u0 = 0;
  if (queued) {
    G.xin = cv.take((int64_t)p->KS * B);
    for (int b = 0; b < p->N; ++b) {
      const int ns = (p->KS - 1) * p->blocks[b].dil.front().dil + 1;
      G.nslots.push_back(ns);
      G.ring.push_back(cv.take((int64_t)ns * B * p->R));
      G.ringp.emplace_back();
      G.nslots_p.emplace_back();
      for (int i = 1; i < p->LPB; ++i) {
        const int nsi = (p->KS - 1) * p->blocks[b].dil[i].dil + 1;
        G.nslots_p.back().push_back(nsi);
        G.ringp.back().push_back(cv.take((int64_t)nsi * B * p->D));
      }
    }
    G.Zrow = cv.take((int64_t)B * p->N * p->Dp);
    G.skiprow = cv.take((int64_t)B * p->Hin);
    G.hrow0 = cv.take((int64_t)B * p->R);
    G.hrow1 = cv.take((int64_t)B * 2 * p->D);
    G.dummy = cv.take((int64_t)B * p->R);
    G.u0 = cv.take(wn_gen_u0_floats(B, p->N, p->D));
    if (p->LPB == 1 && wn_gen_block128_supported(p->R, p->D, p->KS)) G.relay = cv.take(wn_gen_relay128_floats(B, p->N));
    for (size_t i = 0; i + 1 < p->finals.size(); ++i) G.HArow.push_back(cv.take((int64_t)B * p->finals[i].cout));
  }
  G.total = cv.pos;
  return G;
}

__global__ void wn_gen_emit_kernel(const float* samp, int B, float* out, int length, int step, float* xin_slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  out[(int64_t)b * length + step] = samp[b];
  if (xin_slot) xin_slot[b] = samp[b];
}

// sample from the logits rows [B][Cout] (src/model.py:253-255 + sample_waveform)
int sample_rows(wn_plan* p, const float* logits_rows, int B, bool deterministic, uint64_t seed, uint64_t step,
                float* probs_tmp, float* samp, hipStream_t s) {
  const float* pred = logits_rows;
  int rc;
  if (p->c.head == WN_HEAD_CATEGORICAL) {
    rc = wn_launch_softmax(logits_rows, probs_tmp, B, p->Cout, s);     // the model output is probabilities
    if (rc) return rc;
    pred = probs_tmp;
  }
  if (deterministic) return wn_launch_sample_det(pred, B, p->Cout, p->c.num_mixtures, p->c.bits, samp, s);
  return wn_launch_sample_rand(pred, B, p->Cout, p->c.num_mixtures, p->c.bits, p->c.head, seed, step, samp, s);
}

}  // namespace

extern "C" int64_t wn_generate_guard_slot(const wn_plan* p, int32_t B, int32_t queued) {
  if (!p || B < 1) return -1;
  return gen_layout(p, B, queued != 0).guard;
}
extern "C" int64_t wn_generate_workspace_floats(const wn_plan* p, int32_t B, int32_t queued) {
  if (!p || B < 1) return 0;
  return gen_layout(p, B, queued != 0).total;
}

extern "C" int wn_generate(wn_plan* p, const float* params, const float* window, const float* cond, int32_t B,
                           int32_t length, int32_t deterministic, int32_t queued, uint64_t seed, float* out,
                           float* workspace, int64_t ws_floats, void* stream) {
  if (!p || !params || !window || !out || !workspace || B < 1 || length < 0) { wn_set_error("generate: bad arguments"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const int RF = wn_plan_receptive_field(p);
  const GenLayout G = gen_layout(p, B, queued != 0);
  if (ws_floats < G.total) { wn_set_error("generate: workspace too small"); return WN_E_INVALID; }
  if (length == 0) return WN_OK;
  float* pws = workspace + G.prime;
  const WsLayout L = make_layout(p, B, RF, false);
  float* win[2] = {workspace + G.win0, workspace + G.win1};
  float* last = workspace + G.last;
  float* lastp = workspace + G.lastp;
  float* samp = workspace + G.samp;
  WN_HIP_CHECK(hipMemcpyAsync(win[0], window, (int64_t)B * RF * sizeof(float), hipMemcpyDeviceToDevice, s));
  int rc;
  // Range guard (wn_generate_guard_slot): the split-precision kernels cast activations to fp16 hi | lo unscaled, so every
  // kernel that produces one -- priming pass, per-step blocks, the fused chain kernel -- publishes its running max-abs
  // here; the caller reads the float after the call and repeats it with the exact-fp32 kernels when it reached
  // wn_range_limit().  One slot per call: cleared here, only ever raised afterwards.
  float* const gguard = workspace + G.guard;
  WN_HIP_CHECK(hipMemsetAsync(gguard, 0, 2 * sizeof(float), s));   // (and the relay's give-up word behind it)

  if (!queued) {
    // ---- naive sliding window: one full forward over the window per sample (src/model.py:296-305) ----
    for (int step = 0; step < length; ++step) {
      rc = forward_core(p, params, win[step & 1], step == 0, cond, B, RF, false, pws, L, s);
      if (rc) return rc;
      rc = wn_launch_guard_accumulate(pws + L.fwd_absmax, gguard, s);
      if (rc) return rc;
      hipLaunchKernelGGL(wn_gather_last_kernel, dim3((B * p->Cout + 255) / 256), dim3(256), 0, s, pws + L.logits, B, RF, p->Cout, last);
      rc = sample_rows(p, last, B, deterministic != 0, seed, (uint64_t)step, lastp, samp, s);
      if (rc) return rc;
      hipLaunchKernelGGL(wn_gen_shift_kernel, dim3((B * RF + 255) / 256), dim3(256), 0, s, win[step & 1], samp, B, RF,
                         win[(step + 1) & 1], out, length, step);
      WN_HIP_CHECK(hipGetLastError());
    }
    return WN_OK;
  }

  // ---- queued: prime the per-block rings with one forward over the window, then one time step per
  //      sample with rows = utterances; every kernel and every per-row operation order is the one
  //      the sliding window uses, so the results are identical ----
  GenRings R;
  R.xin = workspace + G.xin;
  for (int b = 0; b < p->N; ++b) {
    R.h.push_back(workspace + G.ring[b]);
    R.nslots.push_back(G.nslots[b]);
    R.hp.emplace_back();
    for (int64_t off : G.ringp[b]) R.hp.back().push_back(workspace + off);
    R.nslots_p.push_back(G.nslots_p[b]);
  }
  rc = forward_core(p, params, win[0], true, cond, B, RF, false, pws, L, s, &R);
  if (rc) return rc;
  rc = wn_launch_guard_accumulate(pws + L.fwd_absmax, gguard, s);
  if (rc) return rc;
  hipLaunchKernelGGL(wn_gather_last_kernel, dim3((B * p->Cout + 255) / 256), dim3(256), 0, s, pws + L.logits, B, RF, p->Cout, last);
  rc = sample_rows(p, last, B, deterministic != 0, seed, 0, lastp, samp, s);
  if (rc) return rc;
  // sample 0 is x[RF]; it becomes the network input at time tau = RF
  hipLaunchKernelGGL(wn_gen_emit_kernel, dim3((B + 255) / 256), dim3(256), 0, s, samp, B, out, length, 0,
                     R.xin + (int64_t)(RF % p->KS) * B);
  const float* fragbase = pws + L.frag;
  float* Zrow = workspace + G.Zrow;
  // fused step kernel (input conv + every block in one launch) when the split-precision block kernel
  // is the one the sliding window uses; otherwise the blocks run as separate launches
  // (the fused kernels index rings and rows with 32-bit arithmetic)
  const bool fits32 = (int64_t)(RF + 1) * B * std::max(p->R, p->D) < (1LL << 31) && (int64_t)RF + length < (1LL << 31) &&
                      (int64_t)p->N * B * p->D < (1LL << 31);
  const bool fused_step = p->fused16_ok && p->LPB == 1 && wn_debug_get(1) != 1 && fits32 &&
                          wn_gen_blocks_supported(p->R, p->D, p->KS) && p->N <= wn_gen_chain_max_blocks();
  // the folded form (skip sum and the head's first conv as one contraction, as in forward_core): half the columns
  const bool gfold = fold_ok(p);
  const int skipw = gfold ? p->fold_F0 : p->Sh;
  const int64_t skip_img = gfold ? p->frag16_foldF : p->frag16_skipF;
  const size_t first_final = gfold ? 1 : 0;
  const bool skip_in_chain = fused_step && p->c.use_skip && skip_img >= 0 && wn_gen_skip_fusable(skipw);
  // 128-channel blocks: every block of a step in one launch of wn_gen_chain128_kernel
  const bool chain128 = !fused_step && p->LPB == 1 && p->Dp == p->D && wn_gen_block128_supported(p->R, p->D, p->KS) &&
                        !p->blocks.empty() && p->blocks[0].f16nat >= 0 && p->blocks[0].conv1.frag16 >= 0 && wn_debug_get(1) != 1 && fits32;
  // (its folded skip contraction -- 128 columns -- rides in the same launch)
  const bool skip_in_chain128 = chain128 && gfold && p->c.use_skip && skipw == 128 && skip_img >= 0;
  // ... as a relay over one workgroup per block (wn_gen_relay128_kernel) unless knob 2 asks for the single workgroup
  // (only while every workgroup of a step can be resident at once -- one per CU: a workgroup waits for its predecessor
  // inside the launch; with more workgroups than CUs the chain would still drain as long as they start in index order,
  // which HIP does not promise)
  static int cu_count = 0;
  if (cu_count == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu_count = -1;
  }
  const bool relay128 = chain128 && G.relay >= 0 && wn_debug_get(2) == 0 && (int64_t)((B + 31) / 32) * p->N <= cu_count;
  if (relay128)   // every granule tag starts below the first epoch
    WN_HIP_CHECK(hipMemsetAsync(workspace + G.relay, 0, (size_t)wn_gen_relay128_floats(B, p->N) * sizeof(float), s));
  if ((fused_step || chain128) && (!wnp::ex(p).d_gen || wnp::ex(p).gen_B != B || wnp::ex(p).gen_chain128 != chain128)) {
    wnp::ex(p).gen_chain128 = chain128;
    std::vector<WnGenBlock> tab(p->N);
    for (int b = 0; b < p->N; ++b) {
      const BlockInfo& bi = p->blocks[b];
      WnGenBlock& g = tab[b];
      g.ring_off = G.ring[b];
      g.w16d_off = G.prime + L.frag + (chain128 ? bi.f16nat : bi.dil.back().frag16);
      g.w16r_off = G.prime + L.frag + bi.conv1.frag16;
      g.bias_d_off = p->tensors[bi.dil.back().bias_t].off;
      g.bias_r_off = p->tensors[bi.conv1.bias_t].off;
      g.cb_off = p->c.cond_inputs > 0 ? G.prime + L.cb + (int64_t)b * B * 2 * p->D : -1;
      g.nslots = G.nslots[b];
      g.dilation = bi.dil.back().dil;
    }
    if (wnp::ex(p).d_gen) { (void)hipFree(wnp::ex(p).d_gen); wnp::ex(p).d_gen = nullptr; }
    WN_HIP_CHECK(hipMalloc((void**)&wnp::ex(p).d_gen, tab.size() * sizeof(WnGenBlock)));
    WN_HIP_CHECK(hipMemcpy(wnp::ex(p).d_gen, tab.data(), tab.size() * sizeof(WnGenBlock), hipMemcpyHostToDevice));
    wnp::ex(p).gen_B = B;
    for (int b = 0; b < 3; ++b) wnp::ex(p).gen_blk0[b] = tab[std::min(b, p->N - 1)];
    // conv1 biases at a uniform stride (every block has the same tensors): the chain kernel fetches them without the table
    wnp::ex(p).gen_bias_stride = p->N > 1 ? tab[1].bias_r_off - tab[0].bias_r_off : 1;
    for (int b = 1; b < p->N; ++b)
      if (tab[b].bias_r_off != tab[0].bias_r_off + (int64_t)b * wnp::ex(p).gen_bias_stride) wnp::ex(p).gen_bias_stride = 0;
  }
  const int hc0 = (gfold && p->c.use_skip) ? skipw : p->Hin;
  // the whole head in one launch when every layer is one the split-precision rows GEMM would take
  bool head_fused = fused_step && p->finals.size() > first_final &&
                    (int)(p->finals.size() - first_final) <= WN_GEN_HEAD_MAX && hc0 % 16 == 0 && hc0 <= 256;
  for (size_t i = first_final; i < p->finals.size(); ++i) {
    const ConvInfo& c = p->finals[i];
    head_fused = head_fused && c.frag16 >= 0 && c.cout % 32 == 0 && c.cout >= 64 && c.cout <= 256 && c.cin % 16 == 0 && c.cin <= 256;
  }
  // mixture heads: the same launch with the exact-fp32 last layer (3 x mixtures columns: no split-precision image) and the
  // mixture sampler behind the split-precision layers -- one launch instead of layers + 1 + sampler
  bool head_mix = !head_fused && wn_debug_get(1) != 1 && p->c.head != WN_HEAD_CATEGORICAL && p->c.num_mixtures > 0 &&
                  p->finals.size() >= first_final + 2 && (int)(p->finals.size() - first_final - 1) <= WN_GEN_HEAD_MAX &&
                  hc0 % 16 == 0 && hc0 <= 256 && fits32;
  if (head_mix) {
    for (size_t i = first_final; i + 1 < p->finals.size(); ++i) {
      const ConvInfo& c = p->finals[i];
      head_mix = head_mix && c.frag16 >= 0 && c.cout % 32 == 0 && c.cout >= 64 && c.cout <= 256 && c.cin % 16 == 0 && c.cin <= 256;
    }
    const ConvInfo& cl = p->finals.back();
    head_mix = head_mix && cl.cout == 3 * p->c.num_mixtures && cl.cout <= 32 && cl.cin % 64 == 0 && cl.cin <= 256 && cl.fragF >= 0;
  }
  // the pre kernel's work of step tau + 1 rides in the head launch of step tau
  const bool pre_in_head = head_fused;
  WnGenStepArgs ga;
  memset(&ga, 0, sizeof(ga));
  for (int step = 1; step < length; ++step) {
    const int64_t tau = (int64_t)RF + step - 1;        // time of the newest known sample
    if (fused_step) {
      ga.params = params; ga.ws = workspace; ga.blocks = wnp::ex(p).d_gen; ga.xin = R.xin;
      ga.causal_w = params + p->tensors[p->causal.kernel_t].off;
      ga.causal_b = params + p->tensors[p->causal.bias_t].off;
      ga.u0_off = G.u0;
      for (int b = 0; b < 3; ++b) ga.blk0[b] = wnp::ex(p).gen_blk0[b];
      ga.bias_r_off0 = wnp::ex(p).gen_blk0[0].bias_r_off; ga.bias_r_stride = wnp::ex(p).gen_bias_stride;
      if (skip_in_chain) {
        ga.skip_w16_off = G.prime + L.frag + skip_img;
        ga.skip_bias_off = G.prime + (gfold ? L.bfold : L.bias_sum);
        ga.skiprow_off = G.skiprow; ga.skip_ld = skipw; ga.skip_tiles = skipw / 32;
        ga.skip_act = gfold ? p->c.activation : WN_ACT_LINEAR;
      }
      // the chain kernel raises the guard slot in EVERY step, but only from lanes whose own running max-abs reached the
      // limit (wn_guard_publish_over: no wave reduction, no read of the slot)
      ga.guard = gguard;
      ga.zrow_off = G.Zrow; ga.hrow_off = p->c.use_skip ? -1 : G.hrow0; ga.tau = tau;
      ga.B = B; ga.nblocks = p->N; ga.residual = p->c.use_residual;
      rc = wn_launch_gen_blocks(ga, p->R, p->KS, (pre_in_head && step > 1) ? 2 : 3, s);
      if (rc) return rc;
    } else {
    // input causal conv on [x[tau-(KS-1)], ..., x[tau]]  ->  block 0's ring slot tau  (128-channel chain: inside its launch)
    const bool inconv_in_chain = chain128 && p->KS == 2;
    if (!inconv_in_chain) {
      Gemm g(B, 1, p->R, ceil32(p->R));
      for (int t = 0; t < p->KS; ++t)
        g.seg(R.xin + (int64_t)((tau - (p->KS - 1 - t)) % p->KS) * B, 1, 1, 0,
              fragbase + p->causal.fragF + t * p->causal.fragF_stride);
      rc = g.bias(params + p->tensors[p->causal.bias_t].off).run(R.h[0] + (int64_t)(tau % R.nslots[0]) * B * p->R, p->R, s);
      if (rc) return rc;
    }
    if (chain128) {
      WnGen128Args ca;
      memset(&ca, 0, sizeof(ca));
      ca.params = params; ca.ws = workspace; ca.blocks = wnp::ex(p).d_gen; ca.zrow_off = G.Zrow;
      ca.hrow_off = p->c.use_skip ? -1 : G.hrow0; ca.tau = tau; ca.B = B; ca.nblocks = p->N; ca.residual = p->c.use_residual;
      ca.guard = gguard;
      if (inconv_in_chain) {
        ca.xin = R.xin; ca.causal_w = params + p->tensors[p->causal.kernel_t].off; ca.causal_b = params + p->tensors[p->causal.bias_t].off;
      }
      ca.skip_w16_off = -1;
      if (skip_in_chain128) {
        ca.skip_w16_off = G.prime + L.frag + skip_img; ca.skip_bias_off = G.prime + L.bfold; ca.skiprow_off = G.skiprow;
        ca.skip_act = p->c.activation;
      }
      if (relay128) {
        ca.ntiles = (B + 31) / 32; ca.epoch = (uint32_t)step; ca.relay_off = G.relay;
        ca.tmo = reinterpret_cast<unsigned*>(gguard + 1);
        rc = wn_launch_gen_relay128(ca, s);
      } else {
        rc = wn_launch_gen_chain128(ca, s);
      }
      if (rc) return rc;
    }
    for (int b = 0; b < p->N && !chain128; ++b) {
      BlockPtrs k = block_ptrs(p, b, params, fragbase, B, 1);
      if (p->c.cond_inputs > 0) k.cb = pws + L.cb + (int64_t)b * B * 2 * p->D;
      const int d = p->blocks[b].dil.back().dil;
      BlockBufs f;
      memset(&f, 0, sizeof(f));
      // layers_per_block > 1 (the reference's stated blocker, README.md:16): every dilated conv of the
      // stack has a ring of ITS inputs; the non-gated convs run here, one output row each, and feed the
      // next ring's slot tau
      const float* in_ring = R.h[b];
      int in_ns = R.nslots[b], in_c = p->R;
      for (int i = 0; i + 1 < p->LPB; ++i) {
        const int di = p->blocks[b].dil[i].dil;
        Gemm g(B, 1, p->D, ceil32(p->D));
        for (int t = 0; t < p->KS; ++t)
          g.seg(in_ring + (int64_t)((tau - (int64_t)(p->KS - 1 - t) * di) % in_ns) * B * in_c, in_c, in_c, 0,
                k.Fd[i] + t * k.Fd_stride[i]);
        float* dst = R.hp[b][i] + (int64_t)(tau % R.nslots_p[b][i]) * B * p->D;
        rc = g.bias(k.bd[i]).act(k.act).run(dst, p->D, s);
        if (rc) return rc;
        in_ring = R.hp[b][i]; in_ns = R.nslots_p[b][i]; in_c = p->D;
      }
      for (int t = 0; t < p->KS; ++t)
        f.xt[t] = in_ring + (int64_t)((tau - (int64_t)(p->KS - 1 - t) * d) % in_ns) * B * in_c;
      f.x = f.xt[p->KS - 1];
      if (p->LPB > 1) {
        f.pre_done = true;
        f.res = R.h[b] + (int64_t)(tau % R.nslots[b]) * B * p->R;     // the block input at time tau
      }
      f.U = workspace + G.hrow1;
      f.AG = nullptr;
      f.Z = Zrow + (int64_t)b * B * p->Dp; f.ldz = p->Dp;
      f.O = nullptr;
      f.x_out = (b + 1 < p->N) ? R.h[b + 1] + (int64_t)(tau % R.nslots[b + 1]) * B * p->R
                               : (p->c.use_skip ? workspace + G.dummy : workspace + G.hrow0);
      f.fwd_absmax = gguard;
      rc = block_forward(k, f, s);
      if (rc) return rc;
    }
    }
    const float* hin;
    if (skip_in_chain || skip_in_chain128) {
      hin = workspace + G.skiprow;
    } else if (p->c.use_skip) {
      // utterances are the ROWS of these contractions (no time shift, no per-utterance bias here)
      rc = Gemm(1, B, skipw, ceil32(skipw)).seg_planes(Zrow, p->Dp, (int64_t)B * p->Dp, p->N * p->Dp, gfold ? nullptr : fragbase + p->frag_skipF)
               .w16(skip_img >= 0 ? fragbase + skip_img : nullptr)
               .bias(pws + (gfold ? L.bfold : L.bias_sum)).act(gfold ? p->c.activation : WN_ACT_LINEAR)
               .absmax_fwd(gguard).run(workspace + G.skiprow, skipw, s);
      if (rc) return rc;
      hin = workspace + G.skiprow;
    } else {
      hin = workspace + G.hrow0;
    }
    int hc = hc0;
    bool head_tail = false;
    if (head_fused) {
      WnGenHeadArgs ha;
      memset(&ha, 0, sizeof(ha));
      ha.params = params; ha.ws = workspace; ha.in_off = hin - workspace; ha.in_ld = hc; ha.out_off = G.last;
      ha.nlayers = (int)(p->finals.size() - first_final); ha.B = B;
      ha.guard = gguard;
      for (size_t i = first_final; i < p->finals.size(); ++i) {
        const ConvInfo& c = p->finals[i];
        const size_t l = i - first_final;
        ha.w16_off[l] = G.prime + L.frag + c.frag16; ha.bias_off[l] = p->tensors[c.bias_t].off;
        ha.K[l] = c.cin; ha.N[l] = c.cout;
        ha.act[l] = (i + 1 == p->finals.size()) ? WN_ACT_LINEAR : p->c.activation;
      }
      // categorical heads: the sampling tail and the emit ride in the head launch too
      // (up to 8 utterances = one row per wave of the head workgroup: with more, the rows of a wave run one after the other
      // and the tail kernel's one wave per row finishes sooner -- measured 0.074 vs 0.068 ms per step at B = 32)
      head_tail = p->c.head == WN_HEAD_CATEGORICAL && p->Cout <= 256 && B <= 8;
      if (head_tail) {
        ha.tail = deterministic ? 1 : 2;
        ha.inv_lv = 1.0f / (float)(1 << (p->c.bits - 1));
        ha.seed = seed; ha.offset = (uint64_t)step;
        ha.samp = samp;
        ha.em = WnEmit{out, length, step, R.xin + (int64_t)((tau + 1) % p->KS) * B};
      }
      if (pre_in_head && step + 1 < length) {
        WnGenStepArgs gn = ga;
        gn.tau = tau + 1;
        rc = wn_launch_gen_head_pre(ha, gn, p->R, p->KS, s);
      } else {
        rc = wn_launch_gen_head(ha, s);
      }
      if (rc) return rc;
    } else if (head_mix) {
      WnGenHeadArgs ha;
      memset(&ha, 0, sizeof(ha));
      ha.params = params; ha.ws = workspace; ha.in_off = hin - workspace; ha.in_ld = hc; ha.out_off = G.last;
      ha.nlayers = (int)(p->finals.size() - first_final - 1); ha.B = B;
      ha.guard = gguard;
      for (size_t i = first_final; i + 1 < p->finals.size(); ++i) {
        const ConvInfo& c = p->finals[i];
        const size_t l = i - first_final;
        ha.w16_off[l] = G.prime + L.frag + c.frag16; ha.bias_off[l] = p->tensors[c.bias_t].off;
        ha.K[l] = c.cin; ha.N[l] = c.cout; ha.act[l] = p->c.activation;
      }
      const ConvInfo& cl = p->finals.back();
      ha.f32_w_off = G.prime + L.frag + cl.fragF; ha.f32_bias_off = p->tensors[cl.bias_t].off;
      ha.f32_K = cl.cin; ha.f32_N = cl.cout;
      ha.tail = deterministic ? 3 : 4; ha.mix_M = p->c.num_mixtures; ha.mix_kind = p->c.head;
      ha.seed = seed; ha.offset = (uint64_t)step; ha.samp = samp;
      ha.em = WnEmit{out, length, step, R.xin + (int64_t)((tau + 1) % p->KS) * B};
      rc = wn_launch_gen_head(ha, s);
      if (rc) return rc;
    } else {
    for (size_t i = first_final; i < p->finals.size(); ++i) {
      const ConvInfo& c = p->finals[i];
      const bool lastl = (i + 1 == p->finals.size());
      float* dst = lastl ? last : workspace + G.HArow[i];
      rc = Gemm(1, B, c.cout, ceil32(c.cout)).seg(hin, hc, hc, 0, fragbase + c.fragF)
               .w16(c.frag16 >= 0 ? fragbase + c.frag16 : nullptr)
               .bias(params + p->tensors[c.bias_t].off).act(lastl ? WN_ACT_LINEAR : p->c.activation)
               .absmax_fwd(lastl ? nullptr : gguard).run(dst, c.cout, s);
      if (rc) return rc;
      hin = dst; hc = c.cout;
    }
    }
    if (head_tail || head_mix) {
      // sampled and emitted by the head launch
    } else if (p->c.head == WN_HEAD_CATEGORICAL && deterministic) {
      // softmax + arg max + emit in one launch
      rc = wn_launch_gen_tail_cat_det(last, B, p->Cout, p->c.bits, out, length, step, R.xin + (int64_t)((tau + 1) % p->KS) * B, s);
      if (rc) return rc;
    } else {
      // sampler and emit in one launch (categorical draws straight from the logits: the softmax of
      // wn_softmax_kernel in LDS, the class sample_waveform(softmax(logits)) draws)
      const WnEmit em{out, length, step, R.xin + (int64_t)((tau + 1) % p->KS) * B};
      if (p->c.head == WN_HEAD_CATEGORICAL && !deterministic && wn_sample_from_logits_supported(p->Cout)) {
        rc = wn_launch_sample_rand_cat_logits_emit(last, B, p->Cout, p->c.bits, seed, (uint64_t)step, samp, em, s);
        if (rc) return rc;
      } else if (p->c.head != WN_HEAD_CATEGORICAL) {
        if (deterministic) rc = wn_launch_sample_det_emit(last, B, p->Cout, p->c.num_mixtures, p->c.bits, samp, em, s);
        else rc = wn_launch_sample_rand_emit(last, B, p->Cout, p->c.num_mixtures, p->c.bits, p->c.head, seed, (uint64_t)step, samp, em, s);
        if (rc) return rc;
      } else {
        rc = sample_rows(p, last, B, deterministic != 0, seed, (uint64_t)step, lastp, samp, s);
        if (rc) return rc;
        hipLaunchKernelGGL(wn_gen_emit_kernel, dim3((B + 255) / 256), dim3(256), 0, s, samp, B, out, length, step,
                           R.xin + (int64_t)((tau + 1) % p->KS) * B);
        WN_HIP_CHECK(hipGetLastError());
      }
    }
  }
  return WN_OK;
}
