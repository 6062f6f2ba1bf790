// Plan + orchestration of the WaveNet hot path behind the C-ABI of include/wn_hip.h.
//
// Host-side only (no kernels here): parameter layout in Keras creation order, the table of
// fragment-major weight images, workspace carving, and the launch sequences that restate
//   WaveNet.call          src/model.py:213-239
//   WaveNet.train_step    src/model.py:309-348 (gradient half)
//   WaveNetLayer.call     src/layers.py:178-224
// Nothing here allocates caller-visible memory; the two small device tables a plan owns
// (prep descriptors, tensor table) are immutable launch metadata uploaded on first use.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/wn_hip.h"
#include "wn_kernels.h"

namespace {

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
  int64_t g16r = -1;    // fp16 split image [W_r] alone (used with the precomputed W_s g_skip slice) or -1
  int64_t f16gate = -1; // fp16 split image of the gated conv with row tiles ordered [f f g g] per 64 channels, or -1
  int64_t f16nat = -1;  // fp16 split image of the gated conv in natural row-tile order for the streamed one-kernel forward (R = D = 128), or -1
  int64_t g16uf = -1;   // fp16 split image [W_r | V(b)], V(b) = W_s(b) W_f0 (skip path folded into the first head conv), or -1
  // stacks deeper than 1 (training passes): forward images of the non-gated convs, backward-data images of every conv of
  // the stack; 32-channel outputs are padded to two row tiles (see wn_gemm_rows16_ok)
  int64_t d16F[16], d16B[16];
  BlockInfo() { for (int i = 0; i < 16; ++i) d16F[i] = d16B[i] = -1; }
};

}  // namespace

struct wn_plan {
  wn_config c;
  int KS, R, D, S, N, LPB, Cout, Sh, Hin, Cc, Dp;
  std::vector<int> dilations;
  std::vector<TensorInfo> tensors;
  int64_t nparams = 0;
  ConvInfo causal;
  std::vector<BlockInfo> blocks;
  std::vector<ConvInfo> finals, mapping;
  int64_t frag_skipF = -1;     // A[Sh][N*Dp] image of the folded skip sum
  // all blocks' conv_cond as one layer (when every block has one and 2D % 32 == 0): forward image
  // A[N*2D][Cc], backward image A[Cc][N*2D]; -1 = per-block path
  int64_t frag_condF = -1, frag_condB = -1;
  int64_t frag16_skipF = -1;   // the same as an fp16 split image, or -1
  int64_t frag16_gzs = -1;     // fp16 split image A[N*D][S]: rows b*D.. = W_s of block b (backward of the folded skip sum)
  int64_t frag_floats = 0;
  // skip path folded into the head's first convolution (training passes; see wn_skip_fold_kernel): F0 = its width,
  // forward image A[F0][N*D] of V^T.  prep2 = pieces whose SOURCE is the workspace (the V matrix), not the parameters
  int fold_F0 = 0;
  int64_t frag16_foldF = -1;
  std::vector<WnPrepDesc> prep2;
  WnPrepDesc* d_prep2 = nullptr;
  WnTensorDesc* d_cov_fold = nullptr;
  WnTensorDesc h_cov_fold = {0, 0};     // host copies of the coverage tables (the flattened reduce sizes its grid from them)
  std::vector<WnTensorDesc> h_cov;
  std::vector<WnPrepDesc> prep;
  std::vector<WnTensorDesc> tdesc, kdesc;
  // device copies (lazy)
  WnPrepDesc* d_prep = nullptr;
  WnTensorDesc* d_tdesc = nullptr;
  WnTensorDesc* d_kdesc = nullptr;
  bool fused_ok = false, fused16_ok = false;
  bool deep16_ok = false;      // layers_per_block > 1: split-precision images of the whole stack exist
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
  int jobs_pair_mode = 0;               // 0: one job per tap, 1: staged both-taps job, 2: transposed-read both-taps job
  // the head layers' weight gradients as staged pair jobs (kinds 1..4) on the head's own time split
  int hpair_first[6] = {0, 0, 0, 0, 0, 0}, hpair_count[6] = {0, 0, 0, 0, 0, 0};
  bool jobs_headpairs = false;
  bool jobs_inconvk = false;    // input conv's dW / db from the dedicated reduction kernel, not from jobs
  bool jobs_fold = false;       // tables built for the folded skip path (no conv_skip / first-head-conv entries)
  // side stream: the low-occupancy generic weight-gradient jobs overlap the per-block / skip kernels
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
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

namespace {

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
inline bool m16(int v) { return v > 0 && v % 16 == 0; }
inline bool m32(int v) { return v >= 64 && v % 32 == 0; }

int ensure_device_tables(wn_plan* p) {
  if (p->d_prep) return WN_OK;
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
  return WN_OK;
}

// ------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------
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
  int64_t g_a, g_b;                     // head gradient ping-pong [rows][maxC]
  int64_t g_skipsum;                    // [rows][Hin]
  int64_t g_h0, g_h1;                   // [rows][max(R,D)]
  int64_t g_o;                          // [rows][R]
  int64_t g_p;                          // [rows][D] (depth > 1)
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
  int64_t GZS;                          // [rows][N*D] precomputed W_s g_skip of every block, or 0
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

int64_t slab_need(int B, int T, int K, int N) {
  const int sp = wn_wgrad_choose_splits(B, T, K, N);
  return (int64_t)B * sp * ((int64_t)K * N + N);
}

// The batched weight-gradient path: every dW operand is a whole saved tensor.
// (round 3: stacks deeper than 1 too -- every conv's input and output gradient is kept, the weight gradients of all
//  convs of the step are one launch of the generic job table; knob 17 = 1: the per-call path for them)
bool deferred_wgrad(const wn_plan* p) { return p->LPB == 1 || wn_debug_get(17) != 1; }

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

// Training passes fold the skip path into the head's first convolution when the plan has the images for it
// (fold_F0 > 0: depth-1 blocks with skip convs feeding a head whose first conv is narrower than the skip width) and the
// split-precision kernels run; knob 21 = 1 keeps the reference's two-step form (skip sum, then the head conv).
// stacks deeper than 1 conv, TRAINING passes only: every conv of the stack on the split-precision kernels
// (knob 18 = 1: the exact-fp32 composed kernels, as inference and generation run them)
bool deep16(const wn_plan* p) {
  return p->deep16_ok && wn_debug_get(1) != 1 && wn_debug_get(3) != 1 && wn_debug_get(17) != 1 && wn_debug_get(18) != 1;
}

// The conditioning path works on [B][width] matrices (B = utterances): a rows-GEMM launch per Dense / conv is one wave
// walking its k-steps behind a global round trip each (30-250 us for a few kFLOP).  They run on the small fp32 product
// kernel instead (wn_sgemm_small32_kernel: batched over blocks, Dense epilogue); knob 33 = 1: the rows-GEMM launches.
bool cond_small(const wn_plan* p) {
  if (wn_debug_get(33) == 1 || p->c.cond_inputs <= 0) return false;
  for (int b = 1; b < p->N; ++b) {            // the blocks' conditioning convs must be evenly spaced in the flat buffer
    const int64_t w0 = p->tensors[p->blocks[0].conv_cond.kernel_t].off, w1 = p->tensors[p->blocks[1].conv_cond.kernel_t].off;
    if (!p->blocks[b].has_cond || p->tensors[p->blocks[b].conv_cond.kernel_t].off != w0 + (int64_t)b * (w1 - w0)) return false;
  }
  return p->blocks[0].has_cond;
}

bool fold_ok(const wn_plan* p) {
  // (knob 15 = 1 drops the last block's zero output gradient; the folded g_u product has no one-segment form without it)
  return p->fold_F0 > 0 && wn_debug_get(1) != 1 && wn_debug_get(3) != 1 && wn_debug_get(21) != 1 && wn_debug_get(4) != 1 &&
         wn_debug_get(15) != 1;
}

// the head layers' weight gradients run as staged pair jobs when every final layer has a pair kind (widths 128 / 256)
// in split-precision mode; knob 19 = 1 keeps them on the generic job table
bool head_pairs_ok(const wn_plan* p) {
  if (p->finals.empty() || wn_debug_get(1) == 1 || wn_debug_get(3) == 1 || wn_debug_get(19) == 1) return false;
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
  L.GZS = 0;
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
  if (training && p->drop_rate > 0.f) {
    for (int b = 0; b < p->N; ++b) L.XD.push_back(cv.take(rows * p->R));
    L.gxd = cv.take(rows * p->R);
  }
  if (training) {
    L.g_a = cv.take(rows * maxC);
    L.g_b = cv.take(rows * maxC);
    L.g_skipsum = cv.take(rows * p->Hin);
    L.g_h0 = cv.take(rows * hc);
    L.g_h1 = cv.take(rows * hc);
    L.g_o = cv.take(rows * p->R);
    L.g_p = cv.take(p->LPB > 1 ? 2 * rows * p->D : 0);
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
    if (deferred_wgrad(p)) {
      for (int b = 0; b < p->N; ++b) L.GU.push_back(cv.take(rows * 2 * p->D));
      // depth > 1: gradient w.r.t. the pre-activation output of every non-gated conv (operand of its weight gradient)
      L.GP.assign(p->N, std::vector<int64_t>());
      for (int b = 0; b < p->N; ++b)
        for (int i = 0; i + 1 < p->LPB; ++i) L.GP[b].push_back(cv.take(rows * p->D));
      for (int b = 0; b <= p->N; ++b) L.GH.push_back(cv.take(rows * p->R));
      if (p->S == 0) for (int b = 0; b < p->N; ++b) L.GO.push_back(cv.take(rows * p->R));
      for (size_t i = 0; i < p->finals.size(); ++i) L.GF.push_back(cv.take(rows * p->finals[i].cout));
      L.GZS = p->frag16_gzs >= 0 ? cv.take(rows * p->N * p->D) : 0;
      const int nj = count_jobs(p);
      int sp = (int)((5000 + (int64_t)nj * B - 1) / ((int64_t)nj * B));
      const int maxsp = std::max(1, (T + 255) / 256);
      sp = std::max(1, std::min(sp, maxsp));
      if (wn_debug_get(10) > 0) sp = std::max(1, std::min(wn_debug_get(10), maxsp));   // knob 10: time splits per utterance
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
      // knob 0 = -1: share the blocks' slab and split count; > 0: that many splits
      if (!p->finals.empty() && wn_debug_get(0) >= 0) {
        int njh = 0;
        for (const ConvInfo& c : p->finals) njh += jobs_for(c.cin, c.cout);
        int hs = (int)((1536 + (int64_t)njh * B - 1) / ((int64_t)njh * B));
        // pair jobs: one workgroup per (layer, utterance, time range) -> about one workgroup per CU and layer
        if (head_pairs_ok(p)) hs = std::max(1, (256 + B - 1) / B);
        if (wn_debug_get(0) > 0) hs = wn_debug_get(0);
        hs = std::max(sp, std::min(hs, maxsp));
        if (hs > sp) {
          L.head_base = p->tensors[p->finals.front().kernel_t].off;
          const TensorInfo& last = p->tensors[p->finals.back().bias_t];
          L.head_span = last.off + last.len - L.head_base;
          L.hsplits = hs;
          L.hslab = cv.take((int64_t)B * hs * L.head_span);
        }
      }
    }
  } else {
    L.bslab = 0; L.bsplits = 0;
    L.hslab = 0; L.hsplits = 0; L.head_base = 0; L.head_span = 0;
    L.islab = 0; L.isplits = 0;
    L.g_a = L.g_b = L.g_skipsum = L.g_h0 = L.g_h1 = L.g_o = L.g_p = L.slab = 0;
    L.slab_floats = 0;
  }
  L.total = cv.pos;
  return L;
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------
// one residual block, forward / backward, on explicit pointers (shared by the model
// orchestration and the standalone WaveNetLayer entry points)
// ------------------------------------------------------------------------------------------
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
  const float* G16r;                      // [W_r] alone
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

int block_forward(const BlockPtrs& k, const BlockBufs& f, hipStream_t s) {
  const int64_t rows = (int64_t)k.B * k.T;
  const float* h = f.x;
  int hc = k.Cin;
  int rc;
  if (f.pre_done && k.depth > 1) { h = nullptr; hc = k.D; }
  for (int i = 0; i + 1 < k.depth && !f.pre_done; ++i) {
    const bool i16 = k.F16i[i] != nullptr;            // (training passes of deep stacks: split-precision, see deep16_ptrs)
    // 32 / 64 channels: the streamed kernel's second form with the taps as shifted planes (knob 36 = 1: the rows GEMM)
    if (i16 && k.JTi[i] == 2 && k.KS <= 4 && wn_debug_get(36) != 1 && wn_gemm_taps16s_supported(k.D, hc, k.KS, hc, k.D) &&
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
  // ring and keeps u and z on chip (wn_layer16s.hip); knob 11 = 2 -> the two-contraction form below, = 1 -> exact fp32
  if (k.F16n && k.F16r && k.depth == 1 && k.Cc == 0 && hc == k.R && k.Cin == k.R && f.ldz % 4 == 0 && wn_debug_get(1) != 1 &&
      wn_debug_get(11) == 0 && wn_layer_fwd_s128_supported(k.R, k.D, k.KS) && (int64_t)rows * k.R * 4 < ((int64_t)1 << 32)) {
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
    // requested up front instead of the streamed pipeline (wn_gen128.hip, same arithmetic; knob 34 = 1: the streamed kernel)
    if (k.T == 1 && f.xt[0] && f.xt[1] && !f.AG && wn_debug_get(34) != 1 && wn_gen_block128_supported(k.R, k.D, k.KS))   // (34 = 2 too)
      return wn_launch_gen_block128(a, s);
    return wn_launch_layer_fwd_s128(a, s);
  }
  // the same blocks as [gated conv + gate] -> [1x1 + residual], two split-precision contractions, ahead of the exact-fp32
  // one-kernel forward   (knob 11 = 1 disables it)
  if (k.F16g && k.F16r && k.Cc == 0 && !k.cb && !f.O && hc == k.R && f.ldz % 4 == 0 && wn_debug_get(1) != 1 &&
      wn_debug_get(11) != 1) {
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
  const float* gzs; int ld_gzs;                    // precomputed W_s g_skip slice of this block, or null
  const float* g_fold; int fold_F0; const float* am_gfold;   // folded skip path: dL/da of the first head conv [rows][F0] replaces g_skip
  float drop_rate; uint32_t drop_key; float* g_xd;  // dropout on the block input: mask the conv-path gradient
};

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
    const bool use_gzs = g.gzs && g_o && k.G16r && g.am_gu && g.am_gxout && k.Cc == 0;
    const bool use_fold = g.g_fold && g_o && k.G16uf && g.am_gu && g.am_gxout && g.am_gfold && k.Cc == 0;
    if (g_o) gm.seg(g_o, k.R, k.R, 0, (use_fold || pad_u) ? nullptr : k.Br_);
    if (use_fold) gm.seg(g.g_fold, g.fold_F0, g.fold_F0, 0, nullptr).w16(k.G16uf).absmax(g.am_gxout, g.am_gfold, g.am_gu);
    else if (use_gzs) gm.addc(g.gzs, g.ld_gzs).w16(k.G16r).absmax(g.am_gxout, nullptr, g.am_gu);
    else if (k.S > 0 && g.g_skip) gm.seg(g.g_skip, k.S, k.S, 0, pad_u ? nullptr : k.Bs);
    if (use_gzs || use_fold) {
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
    // instantiation with the taps as negatively shifted planes (knob 36 = 1: the rows GEMM)
    if (b16 && i > 0 && k.JTb[i] == 2 && k.KS <= 4 && wn_debug_get(36) != 1 && wn_gemm_taps16s_supported(hc, gc, k.KS, gc, hc) &&
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

}  // namespace

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
  if (p->LPB == 1 && p->c.use_skip && p->S > 0 && m32(p->D) && m16(p->S) && m16(p->R) && p->Dp == p->D) {
    // W_s g_skip for ALL blocks in one contraction (g_skip is shared): image rows b*D.. = W_s of block b
    p->frag16_gzs = new_image16(p, p->N * p->D, p->S);
    for (int b = 0; b < p->N; ++b) {
      BlockInfo& bi = p->blocks[b];
      add_piece16(p, p->frag16_gzs, p->D, p->tensors[bi.conv_skip.kernel_t].off, p->S, p->S, 0, 0, b * (p->D / 32),
                  p->N * p->D / 32);
      bi.g16r = new_image16(p, p->D, p->R);
      add_piece16(p, bi.g16r, p->D, p->tensors[bi.conv1.kernel_t].off, p->R, p->R, 0, 0);
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
  return p;
}

extern "C" void wn_plan_destroy(wn_plan* p) {
  if (!p) return;
  if (p->d_prep) (void)hipFree(p->d_prep);
  if (p->d_tdesc) (void)hipFree(p->d_tdesc);
  if (p->d_kdesc) (void)hipFree(p->d_kdesc);
  if (p->d_prep2) (void)hipFree(p->d_prep2);
  if (p->d_cov_fold) (void)hipFree(p->d_cov_fold);
  for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->stack_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->foldprep_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->phase_ev) if (e) (void)hipEventDestroy(e);
  if (p->d_jobs) (void)hipFree(p->d_jobs);
  if (p->d_cov) (void)hipFree(p->d_cov);
  if (p->d_wgl) (void)hipFree(p->d_wgl);
  if (p->d_wgli) (void)hipFree(p->d_wgli);
  if (p->d_pairs) (void)hipFree(p->d_pairs);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->d_gen) (void)hipFree(p->d_gen);
  delete p;
}

// ---- Dropout(rate) on every block input in training mode (src/layers.py:108-111, 195-196) ----
extern "C" int wn_plan_set_dropout(wn_plan* p, float rate, uint64_t seed, uint64_t step) {
  if (!p || rate < 0.f || rate >= 1.f) { wn_set_error("Dropout must be between 0 and 1."); return WN_E_INVALID; }
  p->drop_rate = rate; p->drop_seed = seed; p->drop_step = step;
  return WN_OK;
}
extern "C" uint32_t wn_dropout_key_for(uint64_t seed, int32_t block, uint64_t step) { return wn_dropout_key(seed, block, step); }

// ---- profiling hook: HIP events around the residual-block forward launches.  When a forward pass runs
//      its N blocks as N back-to-back launches of the fused block kernel (nothing else in between) ONE
//      event pair brackets the chain and counts N launches: a pair per launch adds its own event packets
//      (+5 us per launch on MI355X) to what it measures.  Otherwise one pair per block. ----
extern "C" int wn_prof_enable(wn_plan* p, int32_t max_launches) {
  if (!p) return WN_E_INVALID;
  for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
  p->prof_ev.clear();
  p->prof_cnt.assign(max_launches > 0 ? max_launches : 0, 1);
  p->prof_used = 0;
  p->prof_on = max_launches > 0;
  for (int i = 0; i < 2 * max_launches; ++i) {
    hipEvent_t e;
    WN_HIP_CHECK(hipEventCreate(&e));
    p->prof_ev.push_back(e);
  }
  return WN_OK;
}
// ---- the whole block stack: one event pair per forward pass, first block launch -> end of the folded skip sum ----
extern "C" int wn_stack_prof_enable(wn_plan* p, int32_t max_passes) {
  if (!p) return WN_E_INVALID;
  for (hipEvent_t e : p->stack_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->foldprep_ev) (void)hipEventDestroy(e);
  p->stack_ev.clear();
  p->foldprep_ev.clear();
  p->stack_used = 0;
  p->foldprep_used = 0;
  for (int i = 0; i < 2 * max_passes; ++i) {
    hipEvent_t e;
    WN_HIP_CHECK(hipEventCreate(&e));
    p->stack_ev.push_back(e);
    WN_HIP_CHECK(hipEventCreate(&e));
    p->foldprep_ev.push_back(e);
  }
  return WN_OK;
}
// the same passes' weight-space preparation of the folded skip path (bias sum, V = W_s W_f0, its fp16 images), which
// runs before the first block launch: average per pass, 0 when the plan does not fold.  Call BEFORE wn_stack_prof_read.
extern "C" int wn_stack_prof_read_foldprep(wn_plan* p, int32_t* passes, float* avg_ms) {
  if (!p || !passes || !avg_ms) return WN_E_INVALID;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i + 1 < p->foldprep_used; i += 2) {
    float ms = 0.f;
    WN_HIP_CHECK(hipEventElapsedTime(&ms, p->foldprep_ev[i], p->foldprep_ev[i + 1]));
    tot += ms; ++n;
  }
  *passes = n;
  *avg_ms = n ? (float)(tot / n) : 0.f;
  p->foldprep_used = 0;
  return WN_OK;
}
extern "C" int wn_stack_prof_read(wn_plan* p, int32_t* passes, float* avg_ms) {
  if (!p || !passes || !avg_ms) return WN_E_INVALID;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i + 1 < p->stack_used; i += 2) {
    float ms = 0.f;
    WN_HIP_CHECK(hipEventElapsedTime(&ms, p->stack_ev[i], p->stack_ev[i + 1]));
    tot += ms; ++n;
  }
  *passes = n;
  *avg_ms = n ? (float)(tot / n) : 0.f;
  p->stack_used = 0;
  return WN_OK;
}
// ---- phase marks of wn_train_fwd_bwd (bench.py): events after the forward, the loss, the backward-data
//      chain and at the end; wn_phase_read returns the four durations of the LAST call in milliseconds ----
extern "C" int wn_phase_enable(wn_plan* p, int32_t on) {
  if (!p) return WN_E_INVALID;
  if (on && !p->phase_ev[0])
    for (hipEvent_t& e : p->phase_ev) WN_HIP_CHECK(hipEventCreate(&e));
  p->phase_on = on != 0;
  return WN_OK;
}
extern "C" int wn_phase_read(wn_plan* p, float* ms4) {
  if (!p || !ms4 || !p->phase_ev[0]) return WN_E_INVALID;
  for (int i = 0; i < 4; ++i) WN_HIP_CHECK(hipEventElapsedTime(ms4 + i, p->phase_ev[i], p->phase_ev[i + 1]));
  return WN_OK;
}
// average milliseconds per recorded launch (call after the stream has been synchronised)
extern "C" int wn_prof_read(wn_plan* p, int32_t* launches, float* avg_ms) {
  if (!p || !launches || !avg_ms) return WN_E_INVALID;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i + 1 < p->prof_used; i += 2) {
    float ms = 0.f;
    WN_HIP_CHECK(hipEventElapsedTime(&ms, p->prof_ev[i], p->prof_ev[i + 1]));
    tot += ms; n += p->prof_cnt[i / 2];
  }
  *launches = n;
  *avg_ms = n ? (float)(tot / n) : 0.f;
  p->prof_used = 0;
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
  else if (!exact && !p->blocks.empty() && p->blocks[0].f16nat >= 0 && wn_debug_get(11) == 0)
    fwd = "fused split-precision block kernel, weights streamed through an LDS ring (wn_layer_fwd_s128_kernel)";
  else if (!exact && !p->blocks.empty() && p->blocks[0].f16gate >= 0 && wn_debug_get(11) != 1)
    fwd = "two split-precision contractions per block (gated conv + gate, 1x1 + residual)";
  else if (p->fused_ok) fwd = "fused exact-fp32 block kernel (wn_layer_fwd_kernel)";
  else fwd = "composed: rows GEMM -> gate kernel -> rows GEMM";
  const bool fold = fold_ok(p);
  const bool deferred = deferred_wgrad(p);
  const char* bwd;
  if (!deferred) bwd = "per-block composed backward with per-call weight gradients (layers_per_block > 1)";
  else if (p->LPB > 1) bwd = "per-block composed backward, one rows contraction per conv of the stack (layers_per_block > 1)";
  else if (fold && p->N >= 2 && wn_bwd_pair_supported(p->R, p->D, p->KS, p->fold_F0) && p->Dp == p->D && wn_debug_get(22) != 1)
    bwd = "two products per launch (wn_bwd_pair_kernel: g_x(b+1) and g_u(b))";
  else if (fold && p->N >= 2 && wn_bwd_s128_supported(p->R, p->D, p->KS, p->fold_F0) && p->Dp == p->D && wn_debug_get(22) != 1)
    bwd = "two products per launch, weights streamed through an LDS ring (wn_bwd_s128_kernel: g_x(b+1) and g_u(b))";
  else if (!exact) bwd = "two split-precision rows contractions per block (g_u with the gate derivative, g_x)";
  else bwd = "two exact-fp32 rows contractions per block";
  const char* wg;
  if (!deferred) wg = "per-call split-K products (wn_wgrad_kernel) + reduces";
  else if (p->LPB > 1 && deep16(p) && wn_wgrad_layer_supported(p->R, p->D, p->KS) && p->Dp == p->R && wn_debug_get(8) != 1)
    wg = "one workgroup per (conv, utterance, time range): the last conv + 1x1 of a block as for depth 1, inner convs through the kernel's INNER form (wn_wgrad_layer_kernel)";
  else if (p->LPB > 1 && deep16(p)) wg = "generic batched job table, split precision, every conv of every stack in one launch (wn_wgrad_batched_kernel)";
  else if (p->LPB > 1) wg = "generic batched job table in exact fp32, every conv of every stack in one launch (wn_wgrad_batched_kernel)";
  else if (!exact && wn_debug_get(3) != 1 && wn_debug_get(8) != 1 && wn_wgrad_layer_supported(p->R, p->D, p->KS) && p->Dp == p->R)
    wg = "one workgroup per (block, utterance, time range) for dW_d, db_d, dW_r, db_r (wn_wgrad_layer_kernel)";
  else if (!exact && wn_debug_get(3) != 1 && wn_debug_get(13) != 1 && p->KS == 2 && p->R == p->D && p->Dp == p->D &&
           wn_wgrad_pair_kind(p->R, 2 * p->D) == 1 && wn_wgrad_pair_kind(p->D, p->R) == 2)
    wg = wn_debug_get(16) == 1 || wn_debug_get(16) == 2
             ? "staged pair jobs (wn_wgrad_pair_kernel)"
             : (wn_debug_get(16) == 0 && fold && p->fold_F0 == 128 && p->D == 128
                    ? "two jobs per block on transposed LDS reads: both taps of dW_d from one read of du; dW_r together with the "
                      "folded skip path's M = Z^T dL/da from one read of z (wn_wgrad_tr_kernel)"
                    : "two jobs per block on transposed LDS reads: both taps of dW_d from one read of du; dW_r (wn_wgrad_tr_kernel)");
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

// ==========================================================================================
// model forward / backward
// ==========================================================================================
namespace {

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
  if (bi.g16r >= 0) k.G16r = fragbase + bi.g16r;
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

struct FwdCtx {
  WsLayout L;
  float* ws;
  const float* frag;
};

// everything up to the logits; training keeps every activation
// queued generation state: per block a ring of its most recent input rows, [slot][B][R]
struct GenRings {
  float* xin;                  // [KS][B] raw samples
  std::vector<float*> h;       // per block: [nslots_b][B][R] inputs of the block's first dilated conv
  std::vector<int> nslots;
  // layers_per_block > 1: hp[b][i] = [nslots_p[b][i]][B][D] inputs of dilated conv i + 1 (= outputs of conv i)
  std::vector<std::vector<float*>> hp;
  std::vector<std::vector<int>> nslots_p;
};

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

int ring_capture(const float* src, int B, int T, int C, int nslots, float* ring, hipStream_t s) {
  const int64_t n = (int64_t)nslots * B * C;
  hipLaunchKernelGGL(wn_ring_capture_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s,
                     src, B, T, C, nslots, ring);
  WN_HIP_CHECK(hipGetLastError());
  return WN_OK;
}

int forward_core(wn_plan* p, const float* params, const float* x, bool prep, const float* cond, int B,
                 int T, bool training, float* ws, const WsLayout& L, hipStream_t s, const GenRings* rings = nullptr) {
  int rc = ensure_device_tables(p);
  if (rc) return rc;
  const int64_t rows = (int64_t)B * T;
  float* fragbase = ws + L.frag;
  if (prep) {
    rc = wn_launch_prep_table(p->d_prep, (int)p->prep.size(), params, fragbase, s);
    if (rc) return rc;
  }
  const bool fp_prof = prep && fold_ok(p) && p->foldprep_used + 2 <= (int)p->foldprep_ev.size();
  if (fp_prof) (void)hipEventRecord(p->foldprep_ev[p->foldprep_used], s);
  // bias of the folded skip sum = sum over blocks of conv_skip (or conv1) biases
  if (prep && p->c.use_skip) {
    const ConvInfo& c0 = p->blocks[0].has_skip ? p->blocks[0].conv_skip : p->blocks[0].conv1;
    WnVecSumArgs v;
    v.base = params; v.off0 = p->tensors[c0.bias_t].off;
    v.stride = p->N > 1 ? (p->tensors[(p->blocks[1].has_skip ? p->blocks[1].conv_skip : p->blocks[1].conv1).bias_t].off - v.off0) : 0;
    v.count = p->N; v.len = p->Sh; v.out = ws + L.bias_sum;
    rc = wn_launch_vecsum(v, s);
    if (rc) return rc;
  }
  // (inference and the generation priming pass fold too: the queued sampler carries the folded contraction in its chain
  // kernel and must reproduce the sliding window bit for bit)
  const bool fold = fold_ok(p);
  if (fold && prep) {
    const BlockInfo& b0 = p->blocks[0];
    const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.kernel_t].off - p->tensors[b0.conv_skip.kernel_t].off : 0;
    rc = wn_launch_skip_fold(params, p->tensors[b0.conv_skip.kernel_t].off, wst, p->tensors[p->finals[0].kernel_t].off,
                             p->tensors[p->finals[0].bias_t].off, ws + L.bias_sum, p->N, p->D, p->S, p->fold_F0, ws + L.vfold,
                             ws + L.bfold, ws + L.wsall, s);
    if (rc) return rc;
    rc = wn_launch_prep_table(p->d_prep2, (int)p->prep2.size(), ws, fragbase, s, 64);  // sources relative to the workspace
    if (rc) return rc;
  }
  if (fp_prof) { (void)hipEventRecord(p->foldprep_ev[p->foldprep_used + 1], s); p->foldprep_used += 2; }
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
  const bool prof_chain = p->prof_on && !rings && p->LPB == 1 && p->c.cond_inputs == 0 && p->R == p->D &&
                          !(training && p->drop_rate > 0.f) && block_ptrs(p, 0, params, fragbase, B, T).fused;
  const bool stack_prof = !rings && p->stack_used + 2 <= (int)p->stack_ev.size();
  if (stack_prof) (void)hipEventRecord(p->stack_ev[p->stack_used], s);
  for (int b = 0; b < p->N; ++b) {
    BlockPtrs k = block_ptrs(p, b, params, fragbase, B, T);
    if (training && !rings) deep16_ptrs(p, b, fragbase, k);
    if (p->c.cond_inputs > 0) k.cb = ws + L.cb + (int64_t)b * B * 2 * p->D;
    BlockBufs f;
    memset(&f, 0, sizeof(f));
    const int hi = training ? b : (b & 1), ho = training ? b + 1 : ((b + 1) & 1);
    f.x = ws + L.H[hi];
    if (training && p->drop_rate > 0.f) {
      // x = dropout(x) feeds the dilated stack; the residual keeps the original (src/layers.py:192-196)
      rc = wn_launch_dropout(ws + L.H[hi], nullptr, ws + L.XD[b], rows * p->R, p->drop_rate,
                             wn_dropout_key(p->drop_seed, b, p->drop_step), nullptr, s);
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
    const bool prof = p->prof_on && p->prof_used + 2 <= (int)p->prof_ev.size();
    const bool ev0 = prof && (!prof_chain || b == 0), ev1 = prof && (!prof_chain || b == p->N - 1);
    if (ev0) (void)hipEventRecord(p->prof_ev[p->prof_used], s);
    rc = block_forward(k, f, s);
    if (rc) return rc;
    if (ev1) {
      (void)hipEventRecord(p->prof_ev[p->prof_used + 1], s);
      p->prof_cnt[p->prof_used / 2] = prof_chain ? p->N : 1;
      p->prof_used += 2;
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
  // skip sum folded into one contraction over all blocks' gated activations (src/model.py:235-236
  // with src/layers.py:216-219), or the last block output when use_skip is False
  const float* hin;
  size_t first_final = 0;
  if (fold) {
    // a = act(sum_b V(b)^T z_b + b'): the skip sum and the head's first conv in ONE contraction with F0 output columns
    const ConvInfo& c0 = p->finals[0];
    // streamed kernel, second form (wn_gemm16s.hip: bit-identical results; knob 31 = 1: wn_gemm_rows16_kernel)
    // (the streamed form indexes rows with 32-bit byte offsets: beyond 4 GiB per plane the rows GEMM takes over)
    if (wn_debug_get(31) != 1 && p->Dp == p->D && wn_gemm_planes16s_supported(c0.cout, p->D, p->N, p->Dp, c0.cout) &&
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
  if (stack_prof) { (void)hipEventRecord(p->stack_ev[p->stack_used + 1], s); p->stack_used += 2; }
  // head, src/model.py:105-119,237-238: conv -> activation, last conv linear (softmax applied later)
  int hc = fold ? p->finals[0].cout : p->Hin;
  for (size_t i = first_final; i < p->finals.size(); ++i) {
    const ConvInfo& c = p->finals[i];
    const bool last = (i + 1 == p->finals.size());
    float* dst = last ? ws + L.logits : ws + L.HA[i];
    // 128 / 256 output columns: the streamed kernel's second form (wn_gemm16s.hip; the operand is one "plane"); same
    // products in the same order as the rows GEMM below (knob 31 = 1)
    if (c.frag16 >= 0 && wn_debug_get(1) != 1 && wn_debug_get(31) != 1 && wn_gemm_planes16s_supported(c.cout, hc, 1, hc, c.cout) &&
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

__global__ void wn_shift_split_kernel(const float* x_full, int B, int T, float* inputs, float* y_true) {
  const int64_t n = (int64_t)B * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / T, t = i % T;
    inputs[i] = x_full[b * (T + 1) + t];        // x[:, :-1]   src/model.py:321
    y_true[i] = x_full[b * (T + 1) + t + 1];    // x[:, 1:]    src/model.py:319
  }
}

int loss_stage(wn_plan* p, int B, int T, int global_batch, bool want_grad, float* ws, const WsLayout& L,
               float* loss_out, float* absmax_out, hipStream_t s) {
  const int64_t rows = (int64_t)B * T;
  const float gscale = 1.0f / (float)global_batch;     // compute_average_loss, src/model.py:328-329
  int rc;
  // deferred weight gradients read d loss / d logits from GF.back(): written there directly (no 131 MB copy)
  float* g_logits = want_grad ? ((deferred_wgrad(p) && !L.GF.empty()) ? ws + L.GF.back() : ws + L.g_a) : nullptr;
  if (p->c.head == WN_HEAD_CATEGORICAL) {
    rc = wn_launch_quantize(ws + L.yt, reinterpret_cast<int32_t*>(ws + L.target), rows, p->c.bits, s);
    if (rc) return rc;
    // an armed step sample (wn_plan_arm_step_sample) rides in the loss kernel when the row fits its registers
    float* so = nullptr;
    if (want_grad && p->step_sample && !p->step_sample_det && p->Cout <= 256) { so = p->step_sample; p->step_sample = nullptr; }
    rc = wn_launch_cat_loss(ws + L.logits, reinterpret_cast<const int32_t*>(ws + L.target), rows, p->Cout,
                            gscale, ws + L.loss_rows, g_logits, absmax_out, s, so, p->c.bits, p->step_sample_seed,
                            p->step_sample_off);
  } else {
    rc = wn_launch_mix_loss(ws + L.logits, ws + L.yt, rows, p->c.num_mixtures, p->c.bits,
                            p->c.head == WN_HEAD_LOGISTIC ? 1 : 2, gscale, ws + L.loss_rows, g_logits, absmax_out, s);
  }
  if (rc) return rc;
  return wn_launch_sum(ws + L.loss_rows, rows, gscale, loss_out, ws + L.sum_scratch, s);
}

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
  return p->c.use_skip && p->S > 0 && p->Dp == p->D && p->N >= 1 && wn_debug_get(1) != 1 && wn_debug_get(3) != 1 &&
         wn_debug_get(5) != 1 && wn_wgrad_skip_supported(p->D, p->S, p->N * p->D);
}

int ensure_jobs(wn_plan* p, const WsLayout& L, int B, int T) {
  const bool skipk = skip_kernel_ok(p);
  // knob 8 = 1 keeps the per-block weight gradients on the generic job table
  // (stacks deeper than 1 in split-precision training, deep16: the last conv + the 1x1 as for depth 1, every inner conv
  // through the kernel's INNER form)
  const bool layerk = (p->LPB == 1 || deep16(p)) && wn_wgrad_layer_supported(p->R, p->D, p->KS) && p->Dp == p->R && wn_debug_get(1) != 1 &&
                      wn_debug_get(3) != 1 && wn_debug_get(8) != 1;
  // knob 13 = 1 keeps them on the generic job table
  const bool pairk = !layerk && p->LPB == 1 && p->KS == 2 && p->R == p->D && p->Dp == p->D && wn_wgrad_pair_kind(p->R, 2 * p->D) == 1 &&
                     wn_wgrad_pair_kind(p->D, p->R) == 2 && wn_debug_get(1) != 1 && wn_debug_get(3) != 1 &&
                     wn_debug_get(13) != 1;
  // Both taps of a block's gated conv as ONE job (du read once): by the transposed-LDS-read kernel (wn_wgrad_tr.hip,
  // default); knob 16 = 1: one staged job per tap (du read twice); = 2: the staged kernel with both taps (128 accumulator
  // registers beside its staging registers: spills, 22.1 vs 16.4 ms per step at configs[3])
  const int pair_mode = !pairk ? 0 : (wn_debug_get(16) == 1 ? 0 : (wn_debug_get(16) == 2 ? 1 : 2));
  const bool pair_dual = pair_mode != 0;
  // with the transposed-read kernels the folded skip path's M = Z^T dL/da is computed by the dW_r jobs (knob 16 = 3: own kernel)
  // ... and by the per-block kernel of 32 / 64-channel blocks, which stages z anyway (wn_wgrad_layer_kernel<.., true>)
  const bool mfused = pair_mode == 2 && p->D == 128 && fold_ok(p) && p->fold_F0 == 128 && p->Dp == p->D && p->S > 0 &&
                      wn_debug_get(16) != 3;
  // 64- / 32-channel blocks: M as transposed-read jobs over the z of four / eight blocks at a time against one read of
  // dL/da (wn_wgrad_tr kinds 7 / 8; knob 16 = 3: wn_wgrad_skip_kernel)
  const int mtr = (layerk && fold_ok(p) && p->fold_F0 == 128 && p->Dp == p->D && p->S > 0 && wn_debug_get(16) != 3)
                      ? (p->D == 64 ? 7 : (p->D == 32 ? 8 : 0)) : 0;
  const bool fold = fold_ok(p);
  const bool headpairs = head_pairs_ok(p) && L.hsplits > 0;
  // knob 20 = 1 keeps the input conv's weight gradients on the generic job table
  const bool inconvk = L.isplits > 0 && wn_debug_get(20) != 1;
  const bool d16 = deep16(p);
  if (p->d_jobs && p->jobs_B == B && p->jobs_T == T && p->jobs_splits == L.bsplits &&
      p->jobs_drop == (p->drop_rate > 0.f) && p->jobs_skipk == skipk && p->jobs_layerk == layerk &&
      p->jobs_pairk == pairk && p->jobs_pair_mode == pair_mode && p->jobs_mfused == mfused && p->jobs_mtr == mtr && p->jobs_deep16 == d16 && p->jobs_headpairs == headpairs && p->jobs_inconvk == inconvk && p->jobs_fold == fold) return WN_OK;
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
      const int64_t xin0 = p->drop_rate > 0.f ? L.XD[b] : L.H[b];
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
      const int64_t xoff = p->drop_rate > 0.f ? L.XD[b] : L.H[b];
      if (pair_dual) {
        // both taps in one job: x[t - d] | x[t] against ONE read of du (kind 6)
        WnWgPair w;
        memset(&w, 0, sizeof(w));
        w.x_off = xoff; w.g_off = L.GU[b]; w.shift = c.dil;
        w.w_off = p->tensors[c.kernel_t].off;
        w.b_off = p->tensors[c.bias_t].off;
        w.gmax_off = am_GU(b);
        pairs[1].push_back(w);
      } else
      for (int t = 0; t < p->KS; ++t) {
        WnWgPair w;
        memset(&w, 0, sizeof(w));
        w.x_off = xoff; w.g_off = L.GU[b]; w.shift = (p->KS - 1 - t) * c.dil;
        w.w_off = p->tensors[c.kernel_t].off + (int64_t)t * p->R * 2 * p->D;
        w.b_off = t == p->KS - 1 ? p->tensors[c.bias_t].off : -1;
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
      const int64_t xo = i == 0 ? (p->drop_rate > 0.f ? L.XD[b] : L.H[b]) : L.P[b][i - 1];
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
  p->head_first = (int)jobs.size();
  p->cov_head_first = (int)cov.size();
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
  if (p->d_jobs) { (void)hipFree(p->d_jobs); p->d_jobs = nullptr; }
  if (p->d_cov) { (void)hipFree(p->d_cov); p->d_cov = nullptr; }
  WN_HIP_CHECK(hipMalloc((void**)&p->d_jobs, std::max<size_t>(jobs.size(), 1) * sizeof(WnWgJob)));
  if (!jobs.empty()) WN_HIP_CHECK(hipMemcpy(p->d_jobs, jobs.data(), jobs.size() * sizeof(WnWgJob), hipMemcpyHostToDevice));
  WN_HIP_CHECK(hipMalloc((void**)&p->d_cov, cov.size() * sizeof(WnTensorDesc)));
  WN_HIP_CHECK(hipMemcpy(p->d_cov, cov.data(), cov.size() * sizeof(WnTensorDesc), hipMemcpyHostToDevice));
  p->h_cov = cov;
  if (p->d_wgl) { (void)hipFree(p->d_wgl); p->d_wgl = nullptr; }
  if (!wgl.empty()) {
    WN_HIP_CHECK(hipMalloc((void**)&p->d_wgl, wgl.size() * sizeof(WnWgLayer)));
    WN_HIP_CHECK(hipMemcpy(p->d_wgl, wgl.data(), wgl.size() * sizeof(WnWgLayer), hipMemcpyHostToDevice));
  }
  if (p->d_wgli) { (void)hipFree(p->d_wgli); p->d_wgli = nullptr; }
  p->n_wgli = (int)wgli.size();
  if (!wgli.empty()) {
    WN_HIP_CHECK(hipMalloc((void**)&p->d_wgli, wgli.size() * sizeof(WnWgLayer)));
    WN_HIP_CHECK(hipMemcpy(p->d_wgli, wgli.data(), wgli.size() * sizeof(WnWgLayer), hipMemcpyHostToDevice));
  }
  if (p->d_pairs) { (void)hipFree(p->d_pairs); p->d_pairs = nullptr; }
  {
    std::vector<WnWgPair> all;
    for (int kd = 0; kd <= 2; ++kd) {
      p->pair_first[kd] = (int)all.size();
      p->pair_count[kd] = (int)pairs[kd].size();
      all.insert(all.end(), pairs[kd].begin(), pairs[kd].end());
    }
    for (int kd = 1; kd <= 5; ++kd) {
      p->hpair_first[kd] = (int)all.size();
      p->hpair_count[kd] = (int)hpairs[kd].size();
      all.insert(all.end(), hpairs[kd].begin(), hpairs[kd].end());
    }
    if (!all.empty()) {
      WN_HIP_CHECK(hipMalloc((void**)&p->d_pairs, all.size() * sizeof(WnWgPair)));
      WN_HIP_CHECK(hipMemcpy(p->d_pairs, all.data(), all.size() * sizeof(WnWgPair), hipMemcpyHostToDevice));
    }
  }
  p->jobs_layerk = layerk; p->jobs_pairk = pairk; p->jobs_pair_mode = pair_mode; p->jobs_mfused = mfused; p->jobs_mtr = mtr; p->jobs_deep16 = d16; p->jobs_headpairs = headpairs; p->jobs_inconvk = inconvk;
  p->jobs_fold = fold;
  p->njobs = (int)jobs.size(); p->ncov = (int)cov.size();
  p->jobs_B = B; p->jobs_T = T; p->jobs_splits = L.bsplits; p->jobs_drop = p->drop_rate > 0.f;
  p->jobs_skipk = skipk;
  return WN_OK;
}

}  // namespace

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
  hipLaunchKernelGGL(wn_shift_split_kernel, dim3((unsigned)std::min<int64_t>((rows + 255) / 256, 4096)), dim3(256), 0, s,
                     x_full, B, T, inputs, workspace + L.yt);
  WN_HIP_CHECK(hipGetLastError());
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

extern "C" int wn_plan_set_train_phases(wn_plan* p, int32_t phases) {
  if (!p || phases < 1 || phases > 3) { wn_set_error("set_train_phases: 1 (forward + loss), 2 (backward), 3 (both)"); return WN_E_INVALID; }
  p->train_phases = phases;
  return WN_OK;
}

extern "C" int wn_train_fwd_bwd(wn_plan* p, const float* params, const float* x_full, const float* cond,
                                int32_t B, int32_t T, int32_t global_batch, int32_t n_replicas, float* grads,
                                float* loss_out, float* pred_out, float* workspace, int64_t ws_floats,
                                void* stream) {
  if (!p || !params || !x_full || !workspace || !loss_out || !grads || B < 1 || T < 1) { wn_set_error("train_fwd_bwd: bad arguments"); return WN_E_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const WsLayout L = make_layout(p, B, T, true);
  if (ws_floats < L.total) { wn_set_error("train_fwd_bwd: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)L.total); return WN_E_INVALID; }
  if (global_batch <= 0) global_batch = B;
  if (n_replicas <= 0) n_replicas = 1;
  float* ws = workspace;
  const int64_t rows = (int64_t)B * T;
  float* inputs = ws + L.probs;
  // A caller may run the step as two calls (wn_plan_set_train_phases 1, then 2) and queue work of its own in between --
  // the Python mirror reads the loss and the metrics back from there, 4 ms before the step ends.  Everything the second
  // half needs lives in the workspace.
  const int phases = p->train_phases;
  int rc = WN_OK;
  if (phases & 1) {
  hipLaunchKernelGGL(wn_shift_split_kernel, dim3((unsigned)std::min<int64_t>((rows + 255) / 256, 4096)), dim3(256), 0, s,
                     x_full, B, T, inputs, ws + L.yt);
  WN_HIP_CHECK(hipGetLastError());
  if (p->phase_on) (void)hipEventRecord(p->phase_ev[0], s);
  rc = forward_core(p, params, inputs, true, cond, B, T, true, ws, L, s);
  if (rc) return rc;
  if (p->phase_on) (void)hipEventRecord(p->phase_ev[1], s);
  }
  // running max-abs scalars of the gradient tensors (operand scaling of the split-precision GEMMs)
  float* am = ws + L.absmax;
  const int nf = (int)p->finals.size();
  auto am_GF = [&](int i) { return am + i; };
  float* am_gskip = am + nf;
  auto am_GU = [&](int b) { return am + nf + 1 + b; };
  auto am_GH = [&](int b) { return am + nf + 1 + p->N + b; };
  auto am_GP = [&](int b, int i) { return am + nf + 1 + p->N + (p->N + 1) + b * (p->LPB - 1) + i; };
  if (phases & 1) {
  WN_HIP_CHECK(hipMemsetAsync(am, 0, L.n_absmax * sizeof(float), s));
  rc = loss_stage(p, B, T, global_batch, true, ws, L, loss_out, am_GF(nf - 1), s);
  if (rc) return rc;
  if (pred_out) {
    if (p->c.head == WN_HEAD_CATEGORICAL) rc = wn_launch_softmax(ws + L.logits, pred_out, rows, p->Cout, s);
    else rc = hipMemcpyAsync(pred_out, ws + L.logits, rows * p->Cout * sizeof(float), hipMemcpyDeviceToDevice, s) == hipSuccess ? WN_OK : WN_E_HIP;
    if (rc) return rc;
  }
  if (p->step_sample) {
    // sample_waveform(pred) of this step (src/model.py:338) drawn from the logits while they are still hot:
    // no (rows, C) probability tensor is written or re-read
    float* so = p->step_sample;
    p->step_sample = nullptr;
    if (p->c.head == WN_HEAD_CATEGORICAL) {
      if (p->step_sample_det) {
        wn_set_error("step sample: deterministic categorical draws go through wn_sample_waveform");
        return WN_E_UNSUPPORTED;
      }
      rc = wn_launch_sample_rand_cat_logits(ws + L.logits, rows, p->Cout, p->c.bits, p->step_sample_seed, p->step_sample_off, so, s);
    } else {
      // mixture heads: the model output IS the logits tensor
      if (p->step_sample_det) rc = wn_launch_sample_det(ws + L.logits, rows, p->Cout, p->c.num_mixtures, p->c.bits, so, s);
      else rc = wn_launch_sample_rand(ws + L.logits, rows, p->Cout, p->c.num_mixtures, p->c.bits, p->c.head, p->step_sample_seed, p->step_sample_off, so, s);
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
  rc = wn_launch_guard_flag(ws + L.fwd_absmax, WN_RANGE_LIMIT * (p->drop_rate > 0.f ? 1.f - p->drop_rate : 1.f),
                            wn_debug_get(1) != 1, loss_out + 2, s);
  if (rc) return rc;
  if (p->phase_on) (void)hipEventRecord(p->phase_ev[2], s);
  }
  if (!(phases & 2)) return WN_OK;
  const float* fragbase = ws + L.frag;
  float* slab = ws + L.slab;

  const bool defer = deferred_wgrad(p);
  const float* mlast = nullptr;
  if (p->c.cond_inputs > 0) {
    mlast = p->mapping.empty() ? cond : ws + L.M.back();
    rc = wn_launch_fill(ws + L.g_m0, 0.f, (int64_t)B * p->Cc, s);
    if (rc) return rc;
  }
  auto cond_block_bwd = [&](const BlockInfo& bi) -> int {
    // conv_cond on the time-invariant mapped condition: dW_c = m^T dcb, db_c = sum_b dcb,
    // g_m += dcb W_c^T
    const ConvInfo& c = bi.conv_cond;
    int r = wgrad(mlast, p->Cc, p->Cc, 0, ws + L.dcb, 2 * p->D, 2 * p->D, 1, B, grads + p->tensors[c.kernel_t].off,
                  grads + p->tensors[c.bias_t].off, nullptr, slab, s);
    if (r) return r;
    return Gemm(1, B, p->Cc, ceil32(p->Cc)).seg(ws + L.dcb, 2 * p->D, 2 * p->D, 0, fragbase + c.fragB)
        .addc(ws + L.g_m0, p->Cc).run(ws + L.g_m0, p->Cc, s);
  };

  // conditioning of all blocks as one layer: the per-utterance sums of d u come out of the weight-gradient
  // slab afterwards instead of 2 column-sum launches + 3 tiny products per block (knob 14 = 1: per block)
  const bool cond_batched = defer && p->frag_condB >= 0 && wn_debug_get(14) != 1;
  if (defer) {
    // ================= data gradients now, every weight gradient in one batched launch =================
    rc = ensure_jobs(p, L, B, T);
    if (rc) return rc;
    // (the loss stage wrote d loss / d logits straight into GF.back(), the last final layer's g)
    const bool fold = fold_ok(p);                  // (the forward pass of this call made the same decision)
    float* head_out = p->c.use_skip ? ws + L.g_skipsum : ws + L.GH[p->N];
    for (int i = (int)p->finals.size() - 1; i >= (fold ? 1 : 0); --i) {
      const ConvInfo& c = p->finals[i];
      float* dst = (i == 0) ? head_out : ws + L.GF[i - 1];
      // 128 / 256 input channels: the streamed kernel's second form in its backward-data instantiation (knob 31 = 1: rows GEMM)
      if (c.frag16B >= 0 && wn_debug_get(1) != 1 && wn_debug_get(31) != 1 && wn_gemm_planes16s_supported(c.cin, c.cout, 1, c.cout, c.cin) &&
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
    // folded skip path: the gradient of the skip sum is never formed; the blocks contract dL/da = GF[0] with V(b)
    const float* g_skip = (p->c.use_skip && !fold) ? ws + L.g_skipsum : nullptr;
    if (p->c.use_skip) {
      rc = wn_launch_fill(ws + L.GH[p->N], 0.f, rows * p->R, s);   // nothing flows into the last block output
      if (rc) return rc;
    }
    // W_s g_skip of every block in ONE contraction: each block then reads its D-column slice (33 MB at
    // configs[1]) instead of re-reading g_skip (131 MB).  Measured SLOWER on configs[1] (round 1: 12.6 vs 12.0 ms
    // per step; round 2, also with a block-major [N][rows][D] result so that the slices are contiguous: 8.59 vs
    // 7.68 ms -- the K = 256, N = 1920 product costs ~1.5 ms, far more than the re-reads it saves), so it is
    // opt-in (knob 4 = 1).
    bool have_gzs = false;
    if (L.GZS > 0 && p->frag16_gzs >= 0 && wn_debug_get(1) != 1 && wn_debug_get(4) == 1) {
      Gemm gz(B, T, p->N * p->D, p->N * p->D / 32);
      gz.seg(g_skip, p->S, p->S, 0, nullptr).w16(fragbase + p->frag16_gzs).absmax(am_gskip, nullptr, nullptr);
      gz.a.y = ws + L.GZS; gz.a.ldy = p->N * p->D; gz.a.vec_out = 1;
      if (wn_gemm_rows16_ok(gz.a)) {
        rc = gz.run(ws + L.GZS, p->N * p->D, s);
        if (rc) return rc;
        have_gzs = true;
      }
    }
    // Two products per launch (wn_bwd_pair.hip): g_x(b+1) and, from it in registers, g_u(b).  The chain is then
    //   g_u(N-1) | { g_x(b+1), g_u(b) } for b = N-2 .. 0 | g_x(0)   = N + 1 launches instead of 2 N.   knob 22 = 1: two launches per block
    const bool pairk = fold && p->N >= 2 && p->drop_rate == 0.f && p->c.use_residual && (p->c.cond_inputs == 0 || cond_batched) && !have_gzs &&
                       (wn_bwd_pair_supported(p->R, p->D, p->KS, p->fold_F0) || wn_bwd_s128_supported(p->R, p->D, p->KS, p->fold_F0)) &&
                       p->Dp == p->D && wn_debug_get(22) != 1 &&
                       wn_debug_get(15) != 1 &&
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
      f.x = (p->drop_rate > 0.f) ? ws + L.XD[b] : ws + L.H[b];
      for (int i = 0; i + 1 < p->LPB; ++i) f.P[i] = ws + L.P[b][i];
      f.AG = ws + L.AG[b];
      f.Z = ws + L.Z + (int64_t)b * rows * p->Dp; f.ldz = p->Dp;
      BlockGrads bg;
      memset(&bg, 0, sizeof(bg));
      bg.defer = true;
      for (int i = 0; i + 1 < p->LPB; ++i) bg.g_pi[i] = ws + L.GP[b][i];
      if (deep16(p))
        for (int i = 0; i + 1 < p->LPB; ++i) bg.am_gp[i] = am_GP(b, i);
      if (p->drop_rate > 0.f) {
        bg.drop_rate = p->drop_rate; bg.drop_key = wn_dropout_key(p->drop_seed, b, p->drop_step); bg.g_xd = ws + L.gxd;
      }
      // the last block's output gradient is identically zero when the head reads the skip sum
      // (with the skip head nothing flows into the last block's output: GH[N] was zero-filled above.  It is
      //  still passed as a gradient -- unless S == 0, where g_o is assembled from g_skip alone -- so that
      //  the last block runs the same split-precision kernels as the others instead of the fp32 fallback
      //  for the one-segment product)
      bg.g_xout = (p->c.use_skip && b == p->N - 1 && (p->S == 0 || wn_debug_get(15) == 1)) ? nullptr : ws + L.GH[b + 1];
      bg.g_skip = g_skip;
      bg.g_o_tmp = p->S == 0 ? ws + L.GO[b] : nullptr;
      bg.g_u = ws + L.GU[b];
      bg.g_x = pairk ? nullptr : ws + L.GH[b];     // (pairs: the next launch forms g_x of this block)
      bg.dcb = (bi.has_cond && !cond_batched) ? ws + L.dcb : nullptr;
      bg.slab = slab;
      bg.am_gxout = bg.g_xout ? am_GH(b + 1) : nullptr;
      bg.am_gskip = g_skip ? am_gskip : nullptr;
      bg.am_gu = am_GU(b); bg.am_gx = am_GH(b);
      if (have_gzs) { bg.gzs = ws + L.GZS + (int64_t)b * p->D; bg.ld_gzs = p->N * p->D; }
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
    if (p->phase_on) (void)hipEventRecord(p->phase_ev[3], s);      // backward-data chain done
    // the generic jobs left over (input conv, head) are few single-wave jobs: they run beside the
    // per-block and skip kernels on a side stream (disjoint slab regions), joined before the reduce.
    // knob 9 = 1 keeps everything on the caller's stream.
    const bool fork = (p->jobs_layerk || p->jobs_pairk) && wn_debug_get(9) != 1;
    if (fork && !p->side) {
      WN_HIP_CHECK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
      WN_HIP_CHECK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
      WN_HIP_CHECK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    }
    // Whatever happens after the fork, the caller's stream must not run ahead of the side stream's kernels (they
    // read and write the workspace and the gradient slab): an early error return joins through this guard.
    struct SideJoin {
      wn_plan* p; hipStream_t s; bool armed;
      ~SideJoin() {
        if (!armed) return;
        if (hipEventRecord(p->ev_join, p->side) != hipSuccess || hipStreamWaitEvent(s, p->ev_join, 0) != hipSuccess)
          (void)hipStreamSynchronize(p->side);
      }
    } side_join{p, s, false};
    if (fork) {
      WN_HIP_CHECK(hipEventRecord(p->ev_fork, s));
      WN_HIP_CHECK(hipStreamWaitEvent(p->side, p->ev_fork, 0));
      side_join.armed = true;
    }
    if (p->jobs_inconvk) {
      rc = wn_launch_inconv_wgrad(inputs, ws + L.GH[0], B, T, p->R, p->KS, L.isplits, ws + L.islab, (int64_t)(p->KS + 1) * p->R,
                                  0, (int64_t)p->KS * p->R, fork ? p->side : s);
      if (rc) return rc;
    }
    const bool head_own = L.hsplits > 0 && (p->head_first < p->njobs || p->jobs_headpairs);
    rc = wn_launch_wgrad_batched(p->d_jobs, head_own ? p->head_first : p->njobs, ws, ws + L.bslab, p->nparams, B, T, L.bsplits,
                                 fork ? p->side : s, p->LPB > 1 && !p->jobs_deep16);
    if (rc) return rc;
    if (head_own) {
      // job and coverage offsets are offsets into the flat parameter buffer: the compact slab is addressed
      // through a base shifted by -head_base with the head span as its row pitch
      if (p->head_first < p->njobs) {
        rc = wn_launch_wgrad_batched(p->d_jobs + p->head_first, p->njobs - p->head_first, ws, ws + L.hslab - L.head_base,
                                     L.head_span, B, T, L.hsplits, fork ? p->side : s);
        if (rc) return rc;
      }
      if (p->jobs_headpairs)
        for (int kd = 1; kd <= 5; ++kd)
          if (p->hpair_count[kd] > 0) {
            // staged kinds 1 (128 x 256), 3 (256 x 128), 5 (256 x 256 halves) have transposed-read forms (3, 4, 5); knob 16 = 1: staged
            const int trk = kd == 1 ? 3 : (kd == 3 ? 4 : (kd == 5 ? 5 : (kd == 2 ? 2 : 0)));
            if (trk != 0 && p->jobs_pair_mode == 2)        // (with 64-channel blocks the staged head jobs are faster beside the side stream's neighbours)
              rc = wn_launch_wgrad_tr(trk, p->d_pairs + p->hpair_first[kd], p->hpair_count[kd], ws, ws + L.hslab - L.head_base,
                                      L.head_span, B, T, L.hsplits, fork ? p->side : s);
            else
            rc = wn_launch_wgrad_pairs(kd, p->d_pairs + p->hpair_first[kd], p->hpair_count[kd], ws, ws + L.hslab - L.head_base,
                                       L.head_span, B, T, L.hsplits, fork ? p->side : s);
            if (rc) return rc;
          }
    }
    if (fork) WN_HIP_CHECK(hipEventRecord(p->ev_join, p->side));
    for (int kd = 1; kd <= 2; ++kd)
      if (p->jobs_pairk && p->pair_count[kd] > 0) {
        if (p->jobs_pair_mode == 2)                     // transposed-read kernels: both taps of dW_d in one job; dW_r (+ M)
          rc = wn_launch_wgrad_tr(kd == 2 && p->jobs_mfused ? 6 : kd, p->d_pairs + p->pair_first[kd], p->pair_count[kd], ws,
                                  ws + L.bslab, p->nparams, B, T, L.bsplits, s, ws + L.mslab,
                                  (int64_t)p->N * p->D * p->fold_F0 + p->fold_F0);
        else
          rc = wn_launch_wgrad_pairs(kd == 1 && p->jobs_pair_mode == 1 ? 6 : kd, p->d_pairs + p->pair_first[kd],
                                     p->pair_count[kd], ws, ws + L.bslab, p->nparams, B, T, L.bsplits, s);
        if (rc) return rc;
      }
    if (p->jobs_layerk) {
      rc = wn_launch_wgrad_layers(p->d_wgl, p->N, p->R, ws, ws + L.bslab, p->nparams, B, T, L.bsplits, s);
      if (rc) return rc;
      rc = wn_launch_wgrad_layers(p->d_wgli, p->n_wgli, p->R, ws, ws + L.bslab, p->nparams, B, T, L.bsplits, s, 1);
      if (rc) return rc;
    }
    if (fold) {
      // M = Z^T dL/da (N*D x F0) and colsum(dL/da) into their own slab, reduced, then the three small products
      const int F0 = p->fold_F0;
      const int64_t pm = (int64_t)p->N * p->D * F0 + F0;
      if (p->jobs_mtr != 0)
        rc = wn_launch_wgrad_tr(p->jobs_mtr, p->d_pairs + p->pair_first[0], p->pair_count[0], ws, ws + L.mslab, pm, B, T,
                                L.bsplits, s);
      else if (!p->jobs_mfused)
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
      // knob 37 = 1: wn_wgrad_kernel; default: split K in chunks of 128 on the small-product kernel, partial results in the
      // (idle) slab, summed in chunk order
      const int nzk = (nd1 + 127) / 128;
      if (wn_debug_get(37) != 1 && (int64_t)nzk * p->S * F0 <= L.slab_floats) {
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
    } else if (p->jobs_skipk) {
      const BlockInfo& b0 = p->blocks[0];
      const int64_t wst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.kernel_t].off - p->tensors[b0.conv_skip.kernel_t].off : 0;
      const int64_t bst = p->N > 1 ? p->tensors[p->blocks[1].conv_skip.bias_t].off - p->tensors[b0.conv_skip.bias_t].off : 0;
      rc = wn_launch_wgrad_skip(ws + L.Z, p->Dp, ws + L.g_skipsum, p->S, rows, p->N * p->D, p->S, p->D,
                                B * L.bsplits, ws + L.bslab, p->nparams, p->tensors[b0.conv_skip.kernel_t].off, wst,
                                p->tensors[b0.conv_skip.bias_t].off, bst, p->N, am_gskip, s);
      if (rc) return rc;
    }
    if (fork) { WN_HIP_CHECK(hipStreamWaitEvent(s, p->ev_join, 0)); side_join.armed = false; }
    if (cond_batched) {
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
    }
    // coverage entries 0, 1 are the input conv's kernel and bias: from their compact slab when the dedicated kernel ran
    const int cov0 = p->jobs_inconvk ? 2 : 0;
    rc = wn_launch_reduce_table(ws + L.bslab, B * L.bsplits, p->nparams, grads, p->d_cov + cov0,
                                (head_own ? p->cov_head_first : p->ncov) - cov0, s, p->h_cov.data() + cov0);
    if (rc) return rc;
    if (p->jobs_inconvk) {
      rc = wn_launch_reduce_table(ws + L.islab, B * L.isplits, (int64_t)(p->KS + 1) * p->R, grads, p->d_cov, 2, s, p->h_cov.data());
      if (rc) return rc;
    }
    if (head_own) {
      rc = wn_launch_reduce_table(ws + L.hslab - L.head_base, B * L.hsplits, L.head_span, grads, p->d_cov + p->cov_head_first,
                                  p->ncov - p->cov_head_first, s, p->h_cov.data() + p->cov_head_first);
      if (rc) return rc;
    }
    if (!p->c.use_skip && p->S > 0) {
      for (const BlockInfo& bi : p->blocks) {     // unused skip convs: zero gradients
        rc = wn_launch_fill(grads + p->tensors[bi.conv_skip.kernel_t].off, 0.f, p->tensors[bi.conv_skip.kernel_t].len, s);
        if (!rc) rc = wn_launch_fill(grads + p->tensors[bi.conv_skip.bias_t].off, 0.f, p->tensors[bi.conv_skip.bias_t].len, s);
        if (rc) return rc;
      }
    }
  } else {
  // ================= per-call weight gradients (blocks with depth > 1) =================
  // ---- head backward ----
  const float* g = ws + L.g_a;          // d loss / d logits
  float* gnext = ws + L.g_b;
  for (int i = (int)p->finals.size() - 1; i >= 0; --i) {
    const ConvInfo& c = p->finals[i];
    const float* xin = (i == 0) ? (p->c.use_skip ? ws + L.skipsum : ws + L.H[p->N]) : ws + L.HA[i - 1];
    rc = wgrad(xin, c.cin, c.cin, 0, g, c.cout, c.cout, B, T, grads + p->tensors[c.kernel_t].off,
               grads + p->tensors[c.bias_t].off, nullptr, slab, s);
    if (rc) return rc;
    Gemm gm(B, T, c.cin, ceil32(c.cin));
    gm.seg(g, c.cout, c.cout, 0, fragbase + c.fragB);
    float* dst = (i == 0) ? ws + L.g_skipsum : gnext;
    if (i > 0) gm.dact(ws + L.HA[i - 1], c.cin, p->c.activation);
    rc = gm.run(dst, c.cin, s);
    if (rc) return rc;
    if (i > 0) { const float* t = g; g = dst; gnext = const_cast<float*>(t); }
  }
  // ---- blocks, last to first ----
  const float* g_skip = p->c.use_skip ? ws + L.g_skipsum : nullptr;
  const float* g_xout = p->c.use_skip ? nullptr : ws + L.g_skipsum;   // head fed by the last block output
  float* ghbuf[2] = {ws + L.g_h0, ws + L.g_h1};
  for (int b = p->N - 1; b >= 0; --b) {
    BlockPtrs k = block_ptrs(p, b, params, fragbase, B, T);
    const BlockInfo& bi = p->blocks[b];
    BlockBufs f;
    memset(&f, 0, sizeof(f));
    f.x = (p->drop_rate > 0.f) ? ws + L.XD[b] : ws + L.H[b];
    for (int i = 0; i + 1 < p->LPB; ++i) f.P[i] = ws + L.P[b][i];
    f.AG = ws + L.AG[b];
    f.Z = ws + L.Z + (int64_t)b * rows * p->Dp; f.ldz = p->Dp;
    BlockGrads bg;
    memset(&bg, 0, sizeof(bg));
    if (p->drop_rate > 0.f) {
      bg.drop_rate = p->drop_rate; bg.drop_key = wn_dropout_key(p->drop_seed, b, p->drop_step); bg.g_xd = ws + L.gxd;
    }
    bg.g_xout = g_xout; bg.g_skip = g_skip; bg.g_o_tmp = ws + L.g_o; bg.g_u = ws + L.U; bg.g_p = ws + L.g_p;
    bg.g_x = ghbuf[b & 1];
    for (int i = 0; i < p->LPB; ++i) {
      bg.dWd[i] = grads + p->tensors[bi.dil[i].kernel_t].off;
      bg.dbd[i] = grads + p->tensors[bi.dil[i].bias_t].off;
    }
    bg.dWr = grads + p->tensors[bi.conv1.kernel_t].off; bg.dbr = grads + p->tensors[bi.conv1.bias_t].off;
    if (bi.has_skip) { bg.dWs = grads + p->tensors[bi.conv_skip.kernel_t].off; bg.dbs = grads + p->tensors[bi.conv_skip.bias_t].off; }
    bg.dcb = bi.has_cond ? ws + L.dcb : nullptr;
    bg.slab = slab;
    rc = block_backward(k, f, bg, s);
    if (rc) return rc;
    if (bi.has_cond) {
      rc = cond_block_bwd(bi);
      if (rc) return rc;
    }
    g_xout = bg.g_x;
  }
  // ---- input causal conv: only weight gradients ----
  for (int t = 0; t < p->KS; ++t) {
    rc = wgrad(inputs, 1, 1, (p->KS - 1 - t), g_xout, p->R, p->R, B, T,
               grads + p->tensors[p->causal.kernel_t].off + (int64_t)t * p->R,
               (t == p->KS - 1) ? grads + p->tensors[p->causal.bias_t].off : nullptr, nullptr, slab, s);
    if (rc) return rc;
  }
  }
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
  // ---- L2 regulariser, src/model.py:331-334: its gradient (the loss term is formed with the loss, above) ----
  if (p->c.l2_reg_factor > 0.f) {
    rc = wn_launch_axpy_table(grads, params, p->d_kdesc, (int)p->kdesc.size(), 2.0f * p->c.l2_reg_factor / (float)n_replicas, s);
    if (rc) return rc;
  }
  if (p->phase_on) {
    if (!defer) (void)hipEventRecord(p->phase_ev[3], s);             // per-call weight gradients: no separate phase
    (void)hipEventRecord(p->phase_ev[4], s);
  }
  return WN_OK;
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

// forward range guard of an inference call: reads nothing back, only tells where the slot is
extern "C" int64_t wn_plan_range_slot(const wn_plan* p, int32_t B, int32_t T, int32_t training) {
  if (!p || B < 1 || T < 1) return -1;
  return make_layout(p, B, T, training != 0).fwd_absmax;
}
extern "C" float wn_range_limit(void) { return WN_RANGE_LIMIT; }

// ==========================================================================================
// generation: WaveNet.generate / _generation, src/model.py:241-307 (intended semantics:
// SURVEY.md section 9 item 8 -- the reference's bad kwarg / rank bugs are not reproduced)
// ==========================================================================================
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
  G.guard = cv.take(1);
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
  if (wn_debug_get(7) == 1) {                         // knob 7 = 1: print the generation workspace map
    fprintf(stderr, "gen layout: prime=%lld win0=%lld last=%lld xin=%lld Zrow=%lld skiprow=%lld hrow0=%lld hrow1=%lld dummy=%lld u0=%lld total=%lld\n",
            (long long)G.prime, (long long)G.win0, (long long)G.last, (long long)G.xin, (long long)G.Zrow, (long long)G.skiprow,
            (long long)G.hrow0, (long long)G.hrow1, (long long)G.dummy, (long long)G.u0, (long long)G.total);
    for (size_t b = 0; b < G.ring.size(); ++b) fprintf(stderr, "  ring[%zu]=%lld nslots=%d\n", b, (long long)G.ring[b], G.nslots[b]);
  }
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
  WN_HIP_CHECK(hipMemsetAsync(gguard, 0, sizeof(float), s));

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
  const bool fused_step = p->fused16_ok && p->LPB == 1 && wn_debug_get(1) != 1 && wn_debug_get(6) != 1 && fits32 &&
                          wn_gen_blocks_supported(p->R, p->D, p->KS);
  // the folded form (skip sum and the head's first conv as one contraction, as in forward_core): half the columns
  const bool gfold = fold_ok(p);
  const int skipw = gfold ? p->fold_F0 : p->Sh;
  const int64_t skip_img = gfold ? p->frag16_foldF : p->frag16_skipF;
  const size_t first_final = gfold ? 1 : 0;
  const bool skip_in_chain = fused_step && p->c.use_skip && skip_img >= 0 && wn_gen_skip_fusable(skipw) &&
                             wn_debug_get(6) != 2;   // knob 6 = 2: skip contraction as its own launch
  // 128-channel blocks: every block of a step in one launch of wn_gen_chain128_kernel (knob 34 = 1: the streamed forward
  // kernel per block, = 2: wn_gen_block128_kernel per block)
  const bool chain128 = !fused_step && p->LPB == 1 && p->Dp == p->D && wn_gen_block128_supported(p->R, p->D, p->KS) &&
                        !p->blocks.empty() && p->blocks[0].f16nat >= 0 && p->blocks[0].conv1.frag16 >= 0 && wn_debug_get(1) != 1 &&
                        wn_debug_get(11) == 0 && wn_debug_get(34) == 0 && fits32;
  // (its folded skip contraction -- 128 columns -- rides in the same launch; knob 6 = 2: its own launch)
  const bool skip_in_chain128 = chain128 && gfold && p->c.use_skip && skipw == 128 && skip_img >= 0 && wn_debug_get(6) != 2;
  if ((fused_step || chain128) && (!p->d_gen || p->gen_B != B || p->gen_chain128 != chain128)) {
    p->gen_chain128 = chain128;
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
    if (p->d_gen) { (void)hipFree(p->d_gen); p->d_gen = nullptr; }
    WN_HIP_CHECK(hipMalloc((void**)&p->d_gen, tab.size() * sizeof(WnGenBlock)));
    WN_HIP_CHECK(hipMemcpy(p->d_gen, tab.data(), tab.size() * sizeof(WnGenBlock), hipMemcpyHostToDevice));
    p->gen_B = B;
    for (int b = 0; b < 3; ++b) p->gen_blk0[b] = tab[std::min(b, p->N - 1)];
    // conv1 biases at a uniform stride (every block has the same tensors): the chain kernel fetches them without the table
    p->gen_bias_stride = p->N > 1 ? tab[1].bias_r_off - tab[0].bias_r_off : 1;
    for (int b = 1; b < p->N; ++b)
      if (tab[b].bias_r_off != tab[0].bias_r_off + (int64_t)b * p->gen_bias_stride) p->gen_bias_stride = 0;
  }
  const int hc0 = (gfold && p->c.use_skip) ? skipw : p->Hin;
  // the whole head in one launch when every layer is one the split-precision rows GEMM would take
  // (knob 6 = 4: one launch per layer)
  bool head_fused = fused_step && wn_debug_get(6) != 4 && p->finals.size() > first_final &&
                    (int)(p->finals.size() - first_final) <= WN_GEN_HEAD_MAX && hc0 % 16 == 0 && hc0 <= 256;
  for (size_t i = first_final; i < p->finals.size(); ++i) {
    const ConvInfo& c = p->finals[i];
    head_fused = head_fused && c.frag16 >= 0 && c.cout % 32 == 0 && c.cout >= 64 && c.cout <= 256 && c.cin % 16 == 0 && c.cin <= 256;
  }
  // the pre kernel's work of step tau + 1 rides in the head launch of step tau (knob 26 = 1: its own launch)
  const bool pre_in_head = head_fused && wn_debug_get(26) != 1;
  WnGenStepArgs ga;
  memset(&ga, 0, sizeof(ga));
  for (int step = 1; step < length; ++step) {
    const int64_t tau = (int64_t)RF + step - 1;        // time of the newest known sample
    if (fused_step) {
      ga.params = params; ga.ws = workspace; ga.blocks = p->d_gen; ga.xin = R.xin;
      ga.causal_w = params + p->tensors[p->causal.kernel_t].off;
      ga.causal_b = params + p->tensors[p->causal.bias_t].off;
      ga.u0_off = G.u0;
      for (int b = 0; b < 3; ++b) ga.blk0[b] = p->gen_blk0[b];
      ga.bias_r_off0 = p->gen_blk0[0].bias_r_off; ga.bias_r_stride = p->gen_bias_stride;
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
    const bool inconv_in_chain = chain128 && p->KS == 2 && wn_debug_get(26) != 1;
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
      ca.params = params; ca.ws = workspace; ca.blocks = p->d_gen; ca.zrow_off = G.Zrow;
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
      rc = wn_launch_gen_chain128(ca, s);
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
      // categorical heads: the sampling tail and the emit ride in the head launch too (knob 6 = 3: separate kernels)
      // (up to 8 utterances = one row per wave of the head workgroup: with more, the rows of a wave run one after the other
      // and the tail kernel's one wave per row finishes sooner -- measured 0.074 vs 0.068 ms per step at B = 32)
      head_tail = p->c.head == WN_HEAD_CATEGORICAL && p->Cout <= 256 && wn_debug_get(6) != 3 && wn_debug_get(27) != 1 &&
                  (B <= 8 || wn_debug_get(27) == 2);
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
    if (head_tail) {
      // sampled and emitted by the head launch
    } else if (p->c.head == WN_HEAD_CATEGORICAL && deterministic && wn_debug_get(6) != 3) {
      // softmax + arg max + emit in one launch (knob 6 = 3: the three separate kernels)
      rc = wn_launch_gen_tail_cat_det(last, B, p->Cout, p->c.bits, out, length, step, R.xin + (int64_t)((tau + 1) % p->KS) * B, s);
      if (rc) return rc;
    } else {
      // sampler and emit in one launch (categorical draws straight from the logits: the softmax of
      // wn_softmax_kernel in LDS, the class sample_waveform(softmax(logits)) draws)
      const WnEmit em{out, length, step, R.xin + (int64_t)((tau + 1) % p->KS) * B};
      if (p->c.head == WN_HEAD_CATEGORICAL && !deterministic && wn_sample_from_logits_supported(p->Cout) && wn_debug_get(6) != 3) {
        rc = wn_launch_sample_rand_cat_logits_emit(last, B, p->Cout, p->c.bits, seed, (uint64_t)step, samp, em, s);
        if (rc) return rc;
      } else if (p->c.head != WN_HEAD_CATEGORICAL && wn_debug_get(6) != 3) {
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

// ==========================================================================================
// elementwise entry points
// ==========================================================================================
extern "C" int wn_quantize(const float* x, int32_t* idx, int64_t n, int32_t bits, void* stream) {
  return wn_launch_quantize(x, idx, n, bits, (hipStream_t)stream);
}
extern "C" int wn_dequantize(const int32_t* idx, float* x, int64_t n, int32_t bits, void* stream) {
  return wn_launch_dequantize(idx, x, n, bits, (hipStream_t)stream);
}
extern "C" int wn_mulaw(const float* x, float* y, int64_t n, void* stream) { return wn_launch_mulaw(x, y, n, (hipStream_t)stream); }
extern "C" int wn_inv_mulaw(const float* y, float* x, int64_t n, void* stream) { return wn_launch_inv_mulaw(y, x, n, (hipStream_t)stream); }
extern "C" int wn_loss_fn(int32_t head, const void* target, const float* pred, int64_t rows, int32_t C,
                          int32_t num_mixtures, int32_t bits, float* loss_rows, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (head == WN_HEAD_CATEGORICAL)
    return wn_launch_cat_loss_probs(pred, (const int32_t*)target, rows, C, loss_rows, s);
  if (head == WN_HEAD_LOGISTIC || head == WN_HEAD_GAUSSIAN)
    return wn_launch_mix_loss(pred, (const float*)target, rows, num_mixtures, bits, head == WN_HEAD_LOGISTIC ? 1 : 2,
                              1.0f, loss_rows, nullptr, nullptr, s);
  wn_set_error("Loss %d not implemented.", head);
  return WN_E_UNSUPPORTED;
}
extern "C" int wn_sum_squared_error(const float* a, const float* b, int64_t n, float scale, float* out, float* scratch,
                                    void* stream) {
  if (!a || !b || !out || !scratch || n < 1) { wn_set_error("sum_squared_error: bad arguments"); return WN_E_INVALID; }
  return wn_launch_sqdiff_sum(a, b, n, scale, out, scratch, (hipStream_t)stream);
}
extern "C" int wn_plan_arm_step_sample(wn_plan* p, float* sample_out, int32_t deterministic, uint64_t seed, uint64_t offset) {
  if (!p) { wn_set_error("arm_step_sample: null plan"); return WN_E_INVALID; }
  if (sample_out && p->c.head == WN_HEAD_CATEGORICAL && (deterministic || !wn_sample_from_logits_supported(p->Cout))) {
    wn_set_error("arm_step_sample: categorical head needs a stochastic draw over <= 1024 classes");
    return WN_E_UNSUPPORTED;
  }
  p->step_sample = sample_out; p->step_sample_det = deterministic; p->step_sample_seed = seed; p->step_sample_off = offset;
  return WN_OK;
}
extern "C" int wn_sample_waveform(int32_t head, const float* pred, int64_t rows, int32_t C, int32_t num_mixtures,
                                  int32_t bits, int32_t deterministic, uint64_t seed, uint64_t offset, float* out,
                                  void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int M = head == WN_HEAD_CATEGORICAL ? 0 : num_mixtures;
  if (deterministic) return wn_launch_sample_det(pred, rows, C, M, bits, out, s);
  return wn_launch_sample_rand(pred, rows, C, M, bits, head, seed, offset, out, s);
}
