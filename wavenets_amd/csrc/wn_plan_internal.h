// Internal interface of the plan / orchestration translation units behind the C-ABI of include/wn_hip.h
// (wn_plan.hip: plan + workspace layout; wn_block.hip: one residual block; wn_forward.hip: the model's forward pass and
// loss; wn_train.hip: backward pass, weight gradients, optimizer; wn_generate.hip: generation; wn_cabi_ops.hip: the
// elementwise entry points).  Host-side only.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/wn_hip.h"
#include "wn_kernels.h"

namespace wnp {

inline int64_t align64(int64_t v) { return (v + 63) & ~(int64_t)63; }
inline int ceil32(int v) { return (v + 31) / 32; }
inline int ceil8(int v) { return (v + 7) / 8 * 8; }

struct TensorInfo {
  int64_t off, len;
  int ndim;
  int64_t shape[3];
  int is_kernel;
};

// one convolution / dense: raw parameter offsets plus its two weight images
struct ConvInfo {
  int kernel_t = -1, bias_t = -1;   // tensor indices
  int taps = 1, cin = 0, cout = 0, dil = 1;
  int64_t fragF = -1;   // taps images of A[cout][cin]  (forward:  A[n][k] = W[tap][k][n])
  int64_t fragB = -1;   // taps images of A[cin][cout]  (backward: A[k][n] = W[tap][k][n])
  int64_t fragF_stride = 0, fragB_stride = 0;
  int64_t frag16 = -1;  // fp16 hi/lo split forward image (all taps concatenated along k), or -1
  int64_t frag16B = -1; // fp16 split backward-data image A[cin][taps*cout], or -1
};

struct BlockInfo {
  std::vector<ConvInfo> dil;
  ConvInfo conv1, conv_skip, conv_cond;
  bool has_skip = false, has_cond = false;
  int64_t g16u = -1;    // fp16 split image [W_r | W_s] (backward: d z) or -1
  int64_t f16gate = -1; // fp16 split image of the gated conv with row tiles ordered [f f g g] per 64 channels, or -1
  int64_t f16nat = -1;  // fp16 split image of the gated conv in natural row-tile order for the streamed one-kernel forward (R = D = 128), or -1
  int64_t g16uf = -1;   // fp16 split image [W_r | V(b)], V(b) = W_s(b) W_f0 (skip path folded into the first head conv), or -1
  // stacks deeper than 1 (training passes): forward images of the non-gated convs, backward-data images of every conv of
  // the stack; 32-channel outputs are padded to two row tiles (see wn_gemm_rows16_ok)
  int64_t d16F[16], d16B[16];
  BlockInfo() { for (int i = 0; i < 16; ++i) d16F[i] = d16B[i] = -1; }
};

}  // namespace wnp

// The mutable side of a plan: per-caller state of the orchestration ("a wn_plan is immutable and shareable
// across streams").  One per stream of execution: dropout counter, the armed step sample, phase selection, the cached
// weight-gradient job tables and generation block table of the last (B, T) it ran (device memory it owns), its side
// stream and events, its profiling events.  A caller that drives one plan from several host threads / streams creates
// one wn_exec per thread (wn_exec_create) and binds it there (wn_exec_bind: thread-local); unbound callers share the
// plan's own.
struct wn_exec {
  const struct wn_plan* plan = nullptr;
  std::vector<WnTensorDesc> h_cov;      // host copy of the coverage table of the cached jobs (the flattened reduce sizes its grid from it)
  float drop_rate = 0.f;        // Dropout rate applied to every block input in training (src/layers.py:108-111)
  uint64_t drop_seed = 0, drop_step = 0;
  // armed by wn_plan_arm_step_sample: the next training step also draws sample_waveform(pred)
  float* step_sample = nullptr; int step_sample_det = 0; uint64_t step_sample_seed = 0, step_sample_off = 0;
  // batched weight-gradient job table (device), valid for one (B, T) workspace layout
  WnWgJob* d_jobs = nullptr;
  WnTensorDesc* d_cov = nullptr;
  int njobs = 0, ncov = 0, jobs_B = 0, jobs_T = 0, jobs_splits = 0;
  bool jobs_drop = false;
  bool jobs_skipk = false;
  bool jobs_layerk = false;   // per-block dW_d / dW_r come from the layer weight-gradient kernel
  WnWgLayer* d_wgl = nullptr;
  WnWgLayer* d_wgli = nullptr;  // inner convs of deeper stacks (wn_wgrad_layer_kernel<.., INNER>)
  int n_wgli = 0;
  // per-block weight gradients as staged pair jobs (widths the per-block kernel does not cover), by kind
  WnWgPair* d_pairs = nullptr;
  int pair_first[3] = {0, 0, 0}, pair_count[3] = {0, 0, 0};
  bool jobs_pairk = false;
  bool jobs_mfused = false;             // M = Z^T dL/da of the folded skip path rides in the dW_r jobs
  bool jobs_deep16 = false;             // inner gradients of deeper stacks carry max-abs slots (split-precision job kernel)
  int jobs_mtr = 0;                     // ... or is its own transposed-read launch over several blocks' z (kind 7 / 8; pairs index 0)
  // the head layers' weight gradients as staged pair jobs (kinds 1..4) on the head's own time split
  int hpair_first[6] = {0, 0, 0, 0, 0, 0}, hpair_count[6] = {0, 0, 0, 0, 0, 0};
  bool jobs_headpairs = false;
  bool jobs_inconvk = false;    // input conv's dW / db from the dedicated reduction kernel, not from jobs
  bool jobs_fold = false;       // tables built for the folded skip path (no conv_skip / first-head-conv entries)
  // side stream: the low-occupancy generic weight-gradient jobs overlap the per-block / skip kernels
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_ffork = nullptr, ev_fjoin = nullptr;   // forward pass: the fold's weight preparation beside the block chain
  int head_first = 0, cov_head_first = 0;   // job / coverage tables: the head's entries come last
  WnGenBlock* d_gen = nullptr;  // fused generation step: per-block offsets for one batch size
  WnGenBlock gen_blk0[3]{};
  int64_t gen_bias_stride = 0;
  int train_phases = 3;   // wn_plan_set_train_phases: bit 0 forward + loss (+ step sample), bit 1 backward + weight gradients
  int gen_B = 0;
  bool gen_chain128 = false;   // the cached generation table carries the 128-channel chain's images
  // optional HIP-event timing of the fused block-forward launches (bench.py roofline leg)
  std::vector<hipEvent_t> prof_ev;   // pairs (start, stop)
  std::vector<int> prof_cnt;         // launches between the events of pair i
  hipEvent_t phase_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // wn_phase_enable: train-step phase marks
  bool phase_on = false;
  int prof_used = 0;
  bool prof_on = false;
  // wn_stack_prof_enable: event pairs around the whole residual-block stack forward (first block launch -> end of
  // the folded skip contraction): SURVEY.md 8(d)'s t_stack_fwd
  std::vector<hipEvent_t> stack_ev;
  std::vector<hipEvent_t> foldprep_ev;   // pairs around the per-pass weight-space preparation of the folded skip path
  int foldprep_used = 0;
  int stack_used = 0;
};

struct wn_plan {
  wn_config c;
  int KS, R, D, S, N, LPB, Cout, Sh, Hin, Cc, Dp;
  std::vector<int> dilations;
  std::vector<wnp::TensorInfo> tensors;
  int64_t nparams = 0;
  wnp::ConvInfo causal;
  std::vector<wnp::BlockInfo> blocks;
  std::vector<wnp::ConvInfo> finals, mapping;
  int64_t frag_skipF = -1;     // A[Sh][N*Dp] image of the folded skip sum
  // all blocks' conv_cond as one layer (when every block has one and 2D % 32 == 0): forward image
  // A[N*2D][Cc], backward image A[Cc][N*2D]; -1 = per-block path
  int64_t frag_condF = -1, frag_condB = -1;
  int64_t frag16_skipF = -1;   // the same as an fp16 split image, or -1
  int64_t frag_floats = 0;
  // skip path folded into the head's first convolution (training passes; see wn_skip_fold_kernel): F0 = its width,
  // forward image A[F0][N*D] of V^T.  prep2 = pieces whose SOURCE is the workspace (the V matrix), not the parameters
  int fold_F0 = 0;
  int64_t frag16_foldF = -1;
  std::vector<WnPrepDesc> prep2;
  WnPrepDesc* d_prep2 = nullptr;
  WnTensorDesc* d_cov_fold = nullptr;
  WnTensorDesc h_cov_fold = {0, 0};     // host copies of the coverage tables (the flattened reduce sizes its grid from them)
  std::vector<WnPrepDesc> prep;
  std::vector<WnTensorDesc> tdesc, kdesc;
  // device copies (lazy)
  WnPrepDesc* d_prep = nullptr;
  WnTensorDesc* d_tdesc = nullptr;
  WnTensorDesc* d_kdesc = nullptr;
  bool fused_ok = false, fused16_ok = false;
  bool deep16_ok = false;      // layers_per_block > 1: split-precision images of the whole stack exist
  // device copies of the immutable tables above are made on first use (a plan can be created without a GPU): guarded
  // by tables_once, never changed afterwards
  std::mutex tables_mu;
  bool tables_ready = false;
  // Everything that CHANGES while a plan is used lives in a wn_exec (below).  `own` is the execution state of callers that
  // never bind one of their own (wn_exec_bind): created with the plan, destroyed with it.
  struct wn_exec* own = nullptr;
};

namespace wnp {
// the execution state of the calling thread for this plan: the bound one if it belongs to the plan, else the plan's own
wn_exec& ex(const wn_plan* p);
void exec_release(wn_exec* e);


struct Carver {
  int64_t pos = 0;
  int64_t take(int64_t n) { int64_t o = pos; pos = align64(pos + (n > 0 ? n : 0)); return o; }
};

struct WsLayout {
  int64_t frag, bias_sum;
  std::vector<int64_t> H;               // N+1 block inputs/outputs (training) or 2 (inference)
  std::vector<std::vector<int64_t>> P;  // per block: outputs of the non-gated convs (depth > 1)
  int64_t Z;                            // [rows][N*Dp]
  std::vector<int64_t> AG;              // per block [rows][D] saved sigmoid (training)
  std::vector<std::vector<int64_t>> GP; // per block, per non-gated conv: [rows][D] gradient of its pre-activation output
  int64_t U;                            // [rows][2D] scratch of the composed path / g_u
  int64_t O;                            // [rows][R] pre-residual output scratch
  int64_t skipsum;                      // [rows][Hin]
  std::vector<int64_t> HA;              // head activations
  int64_t logits, probs;
  int64_t target, loss_rows, yt;
  int64_t g_skipsum;                    // [rows][Hin]
  int64_t slab, slab_floats;
  std::vector<int64_t> GU, GH, GO, GF;  // deferred-wgrad mode: per-block g_u, g_h (N+1), g_o; per-final g
  int64_t bslab; int bsplits;           // batched slab [B*bsplits][nparams]
  // the head's weight gradients get their own, finer time split: a compact slab [B*hsplits][head_span] over the
  // contiguous parameter range of the final layers (head_base = its first float); 0 splits = share bslab
  int64_t hslab; int hsplits; int64_t head_base, head_span;
  // the input conv's (KS + 1) * R sums have a compact slab of their own too: [B * isplits][(KS + 1) * R] (its kernel
  // and bias are the first two tensors of the flat buffer), so that the 33 MB stream is spread over ~512 workgroups
  int64_t islab; int isplits;
  // folded skip path: V [N*D][F0], b' [F0] (fixed offsets right behind the images), [W_s(all blocks); sum b_s]
  // ([N*D + 1][S]), the slab [B*bsplits][N*D*F0 + F0] of M = Z^T dL/da with the column sums behind it, its reduced
  // form [M; colsum] and Y = [M; colsum] W_f0^T ([N*D + 1][S])
  int64_t vfold, bfold, wsall, mslab, mtot, ytmp;
  std::vector<int64_t> XD;              // dropout: dropped copy of every block input (training)
  int64_t gxd;                          // dropout: scratch for d loss / d (dropped input)
  int64_t absmax; int n_absmax;         // running max-abs scalars: GF[i] | g_skipsum | GU[b] | GH[b] | GP[b][i]
  int64_t fwd_absmax;                   // forward range guard: running max-abs of H[b], skip sum, head activations
  int64_t sum_scratch;
  std::vector<int64_t> M;               // mapping activations [B][w]
  int64_t cb;                           // [N][B][2D]
  int64_t dcb, g_m0, g_m1;
  int64_t cbt;                          // [B][N*2D]: all blocks' conditioning biases / their gradients
  int64_t total;
};

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

struct Gemm {
  WnGemmArgs a;
  const float* w16_ = nullptr;
  const float* am0_ = nullptr;
  const float* am1_ = nullptr;
  float* amo_ = nullptr;
  // split-precision image of ALL segments (concatenated along k); optional max-abs scalars
  Gemm& w16(const float* img) { w16_ = img; return *this; }
  Gemm& absmax(const float* in0, const float* in1, float* out) { am0_ = in0; am1_ = in1; amo_ = out; return *this; }
  // forward range-guard slot: unlike a gradient's scale slot it must also record inf / NaN
  Gemm& absmax_fwd(float* out) { amo_ = out; a.absmax_any = out ? 1 : 0; return *this; }
  Gemm(int B, int T, int N, int JTtot) {
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.N = N; a.JTtot = JTtot; a.act = WN_ACT_LINEAR; a.epi = WN_EPI_PLAIN;
  }
  Gemm& seg(const float* x, int ldx, int K, int shift, const float* frag) {
    WnSeg& s = a.seg[a.nseg++];
    s.x = x; s.ldx = ldx; s.K = K; s.shift = shift; s.frag = frag;
    s.vec = (ldx % 4 == 0 && K % 4 == 0 && al16(x)) ? 1 : 0;
    s.plane_k = 0; s.plane_stride = 0;
    return *this;
  }
  // block-major operand [K / plane_k][rows][plane_k] (the gated activations Z of all blocks)
  Gemm& seg_planes(const float* x, int plane_k, int64_t plane_stride, int K, const float* frag) {
    seg(x, plane_k, K, 0, frag);
    WnSeg& s = a.seg[a.nseg - 1];
    s.plane_k = plane_k; s.plane_stride = plane_stride;
    s.vec = (plane_k % 4 == 0 && plane_stride % 4 == 0 && al16(x)) ? 1 : 0;
    return *this;
  }
  Gemm& bias(const float* b) { a.bias = b; return *this; }
  Gemm& rowbias(const float* b, int ld) { a.rowbias = b; a.ld_rowbias = ld; return *this; }
  Gemm& addc(const float* c, int ld) { a.addc = c; a.ld_addc = ld; return *this; }
  Gemm& act(int act) { a.act = act; return *this; }
  Gemm& dact(const float* ysaved, int ld, int act) { a.epi = WN_EPI_DACT; a.aux = ysaved; a.ld_aux = ld; a.act = act; return *this; }
  Gemm& gate_bwd(const float* g, int ldg, const float* z, int ldz) {
    a.epi = WN_EPI_GATE_BWD; a.aux = g; a.ld_aux = ldg; a.aux2 = z; a.ld_aux2 = ldz;
    return *this;
  }
  Gemm& gate_fwd(float* sig, int ld) { a.epi = WN_EPI_GATE_FWD; a.y2 = sig; a.ld_y2 = ld; return *this; }
  int run(float* y, int ldy, hipStream_t s) {
    a.y = y; a.ldy = ldy;
    bool v = (a.N % 4 == 0) && (ldy % 4 == 0) && al16(y);
    if (a.bias) v = v && al16(a.bias);
    if (a.addc) v = v && (a.ld_addc % 4 == 0) && al16(a.addc);
    if (a.aux) v = v && (a.ld_aux % 4 == 0) && al16(a.aux);
    if (a.aux2) v = v && (a.ld_aux2 % 4 == 0) && al16(a.aux2);
    if (a.y2) v = v && (a.ld_y2 % 4 == 0) && al16(a.y2);
    a.vec_out = v ? 1 : 0;
    // knob 1 = 1 forces the exact-fp32 MFMA kernels
    if (w16_ && wn_debug_get(1) != 1 && wn_gemm_rows16_ok(a)) return wn_launch_gemm_rows16(a, w16_, am0_, am1_, amo_, s);
    if (a.epi == WN_EPI_GATE_FWD) { wn_set_error("gate-forward contraction needs the split-precision kernel (alignment / shape)"); return WN_E_UNSUPPORTED; }
    for (int i = 0; i < a.nseg; ++i)
      if (!a.seg[i].frag) { wn_set_error("contraction without an fp32 weight image needs the split-precision kernel"); return WN_E_UNSUPPORTED; }
    return wn_launch_gemm_rows(a, s);
  }
};

struct BlockPtrs {
  // geometry
  int B, T, KS, R, D, S, Cin, depth, act, residual;
  int dil[16];
  // raw parameters
  const float* Wd[16]; const float* bd[16];
  const float* br; const float* bs; const float* bc;
  // images
  const float* Fd[16]; const float* Bd[16]; int64_t Fd_stride[16], Bd_stride[16];
  const float* Fr; const float* Br_;
  const float* Fs; const float* Bs;
  const float* Fc; const float* Bc;
  int Cc;                   // time-varying condition channels (standalone layer) or 0
  const float* cond;        // [rows][Cc]
  const float* cb;          // [B][2D] per-utterance conditioning bias (model) or null
  bool fused;
  const float* F16d; const float* F16r;   // fp16 split images of the gated conv / conv1, or null
  const float* G16u; const float* G16x;   // fp16 split images of the backward-data products, or null
  const float* G16uf;                     // [W_r | V(b)]: skip path folded into the first head conv (training), or null
  const float* F16g;                      // gated conv, row tiles [f f g g] per 64 channels (composed split-precision forward), or null
  const float* F16n;                      // gated conv, natural row-tile order, for the streamed one-kernel forward (R = D = 128), or null
  // depth > 1, training passes (set by deep16_ptrs): split-precision images of the non-gated convs (forward) and of every
  // conv's backward-data product; JTi / JTb = row tiles of those images (32-wide outputs are padded to 2)
  const float* F16i[16]; const float* G16i[16];
  int JTi[16], JTb[16], JTu;
};

struct BlockBufs {
  const float* x;           // [rows][Cin]
  float* P[16];             // outputs of non-gated convs [rows][D]
  float* U;                 // [rows][2D] scratch
  float* AG;                // [rows][D] saved sigmoid or null
  float* Z; int ldz;        // gated activations
  float* O;                 // [rows][R] pre-residual output or null
  float* x_out;             // [rows][R]
  const float* xt[3];       // queued generation: per-tap input rows (no time shift), or null
  const float* res;         // residual source when it is not x (dropout: x is the dropped copy), or null
  bool pre_done;            // queued generation: the non-gated convs already ran, xt[] are the gated conv's taps
  float* fwd_absmax;        // forward range guard slot (running max-abs of x_out), or null
};

struct BlockGrads {
  const float* g_xout;      // [rows][R] or null (treated as zero)
  const float* g_skip;      // [rows][Sh] or null; Sh = S, or R when S == 0 (skip = pre-residual o)
  float* g_o_tmp;           // [rows][R] scratch (needed when S == 0 and both grads exist)
  float* g_u;               // [rows][2D] scratch
  float* g_p;               // [2][rows][D] scratch (depth > 1), halves used alternately
  float* g_pi[16];          // deferred weight gradients: where the gradient of conv i's pre-activation output is KEPT, or null
  float* g_x;               // [rows][Cin] out (may be null when not needed)
  float* g_cond;            // [rows][Cc] out or null
  float* dWd[16]; float* dbd[16];
  float* dWr; float* dbr; float* dWs; float* dbs; float* dWc; float* dbc;
  float* dcb;               // [B][2D] per-utterance sums of g_u (model conditioning) or null
  float* slab;
  bool defer;               // weight gradients are computed later by the batched job table
  const float* am_gxout; const float* am_gskip;   // running max-abs of g_xout / g_skip (or null)
  float* am_gu; float* am_gx;                      // where to publish max-abs of g_u / g_x (or null)
  float* am_gp[16];                                // ... of the kept inner gradients g_pi[i] (deep stacks in training), or null
  const float* g_fold; int fold_F0; const float* am_gfold;   // folded skip path: dL/da of the first head conv [rows][F0] replaces g_skip
  float drop_rate; uint32_t drop_key; float* g_xd;  // dropout on the block input: mask the conv-path gradient
};

// queued generation state: per block a ring of its most recent input rows, [slot][B][R]
struct GenRings {
  float* xin;                  // [KS][B] raw samples
  std::vector<float*> h;       // per block: [nslots_b][B][R] inputs of the block's first dilated conv
  std::vector<int> nslots;
  // layers_per_block > 1: hp[b][i] = [nslots_p[b][i]][B][D] inputs of dilated conv i + 1 (= outputs of conv i)
  std::vector<std::vector<float*>> hp;
  std::vector<std::vector<int>> nslots_p;
};

inline bool m16(int v) { return v > 0 && v % 16 == 0; }
inline bool m32(int v) { return v >= 64 && v % 32 == 0; }

// ---- wn_plan.hip ----
int ensure_device_tables(wn_plan* p);
WsLayout make_layout(const wn_plan* p, int B, int T, bool training);
bool deep16(const wn_plan* p);
bool cond_small(const wn_plan* p);
bool fold_ok(const wn_plan* p);
bool head_pairs_ok(const wn_plan* p);
int64_t slab_need(int B, int T, int K, int N);
// ---- wn_block.hip ----
int wgrad(const float* x, int ldx, int K, int shift, const float* g, int ldg, int N, int B, int T,
          float* dW, float* db, float* per_batch, float* slab, hipStream_t s);
int block_forward(const BlockPtrs& k, const BlockBufs& f, hipStream_t s);
int block_backward(const BlockPtrs& k, const BlockBufs& f, const BlockGrads& g, hipStream_t s);
// ---- wn_forward.hip ----
// Training passes of a 256-class categorical head: the loss runs as the epilogue of the head's last conv
// (wn_gemm_planes16s_kernel<1, 8, -3>): where that launch puts its results.  forward_core sets `done` when it took the
// fused path (the logits are then never written); loss_stage only sums the row losses.
struct LossFuse {
  const int32_t* target; float gscale; float* loss_rows; float* g_logits; float* absmax_out;
  float* sample_out; float inv_lv; uint64_t seed, offset;
  bool done;
};
bool loss_fusable(const wn_plan* p, int64_t rows);
BlockPtrs block_ptrs(const wn_plan* p, int b, const float* params, const float* fragbase, int B, int T);
void deep16_ptrs(const wn_plan* p, int b, const float* fragbase, BlockPtrs& k);
int forward_core(wn_plan* p, const float* params, const float* x, bool prep, const float* cond, int B,
                 int T, bool training, float* ws, const WsLayout& L, hipStream_t s, const GenRings* rings = nullptr,
                 LossFuse* lf = nullptr);
// inputs = x[:, :-1], y_true = x[:, 1:]  (src/model.py:319-321)
int shift_split(const float* x_full, int B, int T, float* inputs, float* y_true, hipStream_t s);
int loss_stage(wn_plan* p, int B, int T, int global_batch, bool want_grad, float* ws, const WsLayout& L,
               float* loss_out, float* absmax_out, hipStream_t s, bool fused_done = false);

}  // namespace wnp
