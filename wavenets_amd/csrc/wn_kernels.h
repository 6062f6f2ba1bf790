// Internal launcher interface between the kernel translation units and the plan/C-ABI layer.
#pragma once
#include "wn_common.h"

struct WnTensorDesc { int64_t off; int64_t len; };

// ---------------------------------------------------------------- weight preparation
// One descriptor = one dense matrix copied into a fragment-major image (wn_common.h).
// Offsets are in floats relative to the base pointers given at launch, so that a table
// built once at plan creation works with whatever buffers the caller passes per call.
struct WnPrepDesc {
  int64_t src_off;   // into params
  int64_t dst_off;   // into workspace
  int32_t I;         // image rows (output channels of the contraction)
  int32_t KK;        // contraction length
  int32_t ld;        // leading dimension of the source matrix
  int32_t transpose; // 0: A[i][kk] = src[i*ld + kk]   1: A[i][kk] = src[kk*ld + i]
  int32_t q_off;     // first k-quad of this piece inside a larger concatenated image
  int32_t j_off;     // first row tile of this piece inside a larger stacked image
  int32_t JT;        // row tiles of the whole image
  int32_t kind;      // 0: fp32 fragment image; 1: fp16 hi/lo split image for the 32x32x16 MFMA (q_off counts k-steps of 16)
};
int wn_launch_prep_table(const WnPrepDesc* d_table, int n, const float* params, float* ws,
                         hipStream_t s, int gx = 16);
int wn_launch_prep_one(WnPrepDesc d, const float* params, float* ws, hipStream_t s);
// out[i] = sum_j src[offs[j] + i]   (sum of N bias vectors; offsets by value, N <= 64)
struct WnVecSumArgs { const float* base; int64_t off0; int64_t stride; int32_t count; int32_t len; float* out; };
int wn_launch_vecsum(WnVecSumArgs a, hipStream_t s);

// ---------------------------------------------------------------- rows GEMM
#define WN_MAXSEG 4
struct WnSeg {
  const float* x;     // [B*T][ldx] activations, channels contiguous
  const float* frag;  // fragment-major weight image of this K segment (JTtot row tiles)
  int32_t ldx;
  int32_t K;          // valid channels of this segment
  int32_t shift;      // output row t contracts x row t - shift (zero outside [0,T))
  int32_t vec;        // 1: rows are 16-byte aligned and K % 4 == 0 -> vector loads
  int32_t plane_k;    // > 0: the K channels are spread over planes of plane_k channels, plane p at x + p * plane_stride
  int64_t plane_stride;   // (block-major Z: [N][rows][D]); ldx is then the row stride inside a plane
};
struct WnGemmArgs {
  WnSeg seg[WN_MAXSEG];
  int32_t nseg;
  int32_t B, T;
  int32_t N;          // valid output channels
  int32_t JTtot;      // row tiles of the weight images
  const float* bias;      // [N] or null
  const float* rowbias;   // [B][ld_rowbias] or null (per-utterance bias: global conditioning)
  int32_t ld_rowbias;
  const float* addc;      // [B*T][ld_addc] or null
  int32_t ld_addc;
  int32_t act;
  int32_t epi;
  const float* aux;       // WN_EPI_DACT: saved activations; WN_EPI_GATE_BWD: saved sigmoid g [rows][ld_aux]
  int32_t ld_aux;
  const float* aux2;      // WN_EPI_GATE_BWD: gated activation z = tanh * sigmoid [rows][ld_aux2]
  int32_t ld_aux2;
  float* y;
  int32_t ldy;
  int32_t vec_out;        // 1: y/addc/aux rows 16-byte aligned, N % 4 == 0
  float* y2;              // WN_EPI_GATE_FWD: the sigmoid [rows][ld_y2], or null (inference)
  int32_t ld_y2;
  int32_t absmax_any;     // 1: absmax_out is a forward range-guard slot (records inf / NaN too), 0: a gradient's scale slot
};
int wn_launch_gemm_rows(const WnGemmArgs& a, hipStream_t s);
// split-precision variant: w16 = ONE fp16 hi|lo image of all segments' weights concatenated along k
// (kind-1 prep, JTtot row tiles); absmax_* are optional device scalars (running max-abs of the
// B-operand tensors / of the output) used for the exact power-of-two operand scaling
int wn_gemm_rows16_ok(const WnGemmArgs& a);
int wn_launch_gemm_rows16(const WnGemmArgs& a, const float* w16, const float* absmax_in0,
                          const float* absmax_in1, float* absmax_out, hipStream_t s);

// rows contraction over a planar operand with 128 output columns, streamed weights, second form (wn_gemm16s.hip):
// bit-identical to wn_launch_gemm_rows16 on one planar segment with the PLAIN epilogue
struct WnGemmPlanesArgs {
  const float* z;          // plane p, row r: z + p * plane_stride + r * ld
  int64_t plane_stride;    // floats
  int32_t ld, plane_k, nplanes;
  const float* w16;        // fp16 hi|lo image (kind-1 prep, 4 row tiles) of the [nplanes * plane_k][128] weights
  const float* bias;       // [128] or null
  int32_t act;
  float* y; int32_t ldy;
  int32_t N;               // 128
  int32_t B, T;
  float* absmax_out;       // forward range-guard slot (bwd: the result's gradient max-abs slot) or null
  // bwd = 1: backward-data form: operand scaled by the power of two of *absmax_in, epilogue * act'(aux[row][n]) when aux
  // is given (act = the forward activation), no bias
  int32_t bwd;
  const float* absmax_in;
  const float* aux; int32_t ld_aux;
  // nshift == nplanes (<= 4) with plane_stride 0: plane p reads row t - shift[p] of its utterance (the taps of a dilated
  // conv, or of its backward with negative shifts); rows outside [0, T) contribute zero.  Outputs of 32 / 64 channels only.
  int32_t nshift; int32_t shift[4];
  // cat_loss != 0 (N == 256 == classes, forward form): the epilogue is the categorical loss instead of a store of the
  // logits -- Keras sparse CE on clipped probabilities per row (src/model.py:515-516) into loss_rows, d loss / d logits
  // (* gscale) into y, its running max-abs into absmax_out (a gradient slot), and optionally the step's
  // sample_waveform(pred) draw (src/model.py:338,405-411) into sample_out: the (rows, 256) logits never reach HBM
  int32_t cat_loss;
  const int32_t* target;   // [rows] class indices
  float gscale;
  float* loss_rows;        // [rows]
  float* sample_out;       // [rows] or null
  float inv_lv;            // 2 / levels
  uint64_t seed, offset;   // Philox key / counter word of the draw
};
int wn_gemm_planes16s_supported(int N, int plane_k, int nplanes, int ld, int ldy);
int wn_gemm_taps16s_supported(int N, int plane_k, int ntaps, int ld, int ldy);
int wn_launch_gemm_planes16s(const WnGemmPlanesArgs& a, hipStream_t s);

// ---------------------------------------------------------------- weight-gradient GEMM
struct WnWgradArgs {
  const float* x; int32_t ldx; int32_t K; int32_t shift;   // dW[k][n] = sum_rows x[t-shift][k] g[t][n]
  const float* g; int32_t ldg; int32_t N;
  int32_t B, T;
  int32_t splits_per_b;
  float* slab;        // [B*splits_per_b][K][N]
  float* slab_bias;   // [B*splits_per_b][N] or null
};
int wn_wgrad_choose_splits(int B, int T, int K, int N);
int wn_launch_wgrad(const WnWgradArgs& a, hipStream_t s);
// batched form: one job per (K-block, N-block) of some dW; offsets are floats relative to the
// workspace base (operands) and to a slab row / the flat gradient buffer (outputs)
struct WnWgJob {
  int64_t x_off, g_off, out_off, bias_off;   // bias_off < 0: no bias sum from this job
  int64_t gmax_off;                          // workspace offset of the running max-abs of g, or < 0
  int32_t ldx, ldg, K, N, shift, k0, n0, pad_;
};
// dW_d (both taps), db_d, dW_r, db_r of one residual block per workgroup (wn_wgrad_layer.hip)
struct WnWgLayer {
  int64_t x_off, du_off, z_off, go_off;         // workspace offsets: [rows][R], [rows][2D], [rows][ldz], [rows][R]
  int64_t dwd_off, dbd_off, dwr_off, dbr_off;   // offsets inside a slab row (= flat gradient layout)
  int64_t gmax_u_off, gmax_h_off;               // running max-abs of du / go, or < 0
  int32_t dilation, ldz;
};
int wn_wgrad_layer_supported(int R, int D, int KS);
int wn_launch_wgrad_layers(const WnWgLayer* d_layers, int nlayers, int R, float* ws, float* slab, int64_t P, int B,
                           int T, int splits_per_b, hipStream_t s, int inner = 0);
// one tap of a block's weight gradient as a staged workgroup job (wn_wgrad_pair.hip)
struct WnWgPair {
  int64_t x_off, g_off;            // workspace offsets of X [rows][K] and G [rows][N]
  int64_t w_off, b_off;            // dW[k][n] -> slab row + w_off + k * N + n; db -> slab row + b_off (or < 0)
  int64_t gmax_off;                // running max-abs of G, or < 0
  int32_t shift, pad_;             // X row = t - shift
  // transposed-read kernel, two-source form (kind 6): the upper half of G's columns comes from a second tensor and its
  // product goes to a second slab -- dW_r = z^T g_o and M = z^T dL/da (folded skip path) from ONE read of z
  int64_t g2_off, w2_off, b2_off, gmax2_off;
};
int wn_wgrad_pair_kind(int K, int N);
int wn_launch_wgrad_pairs(int kind, const WnWgPair* d_jobs, int njobs, float* ws, float* slab, int64_t P, int B, int T,
                          int splits_per_b, hipStream_t s);
// both taps of a 128 -> 256 kernel-size-2 convolution's weight gradient from one read of G, transposition in the LDS read
// (wn_wgrad_tr.hip); jobs: x_off, g_off, shift = dilation, w_off = tap 0's offset, b_off, gmax_off
int wn_wgrad_tr_kind(int K, int N, int taps);
int wn_launch_wgrad_tr(int kind, const WnWgPair* d_jobs, int njobs, float* ws, float* slab, int64_t P, int B, int T,
                       int splits_per_b, hipStream_t s, float* slab2 = nullptr, int64_t P2 = 0);
// two products of the backward-data chain per launch: g_x(b+1) and, from it in registers, g_u(b) (wn_bwd_pair.hip)
struct WnBwdPairArgs {
  const float* gu_in;    // g_u(b+1) [rows][2D]
  const float* gx_res;   // g_x(b+2) [rows][R]: the residual path's gradient
  const float* gf;       // dL/da of the head's first conv [rows][F0] (folded skip path)
  const float* ag;       // saved sigmoid of block b [rows][D]
  const float* z; int32_t ldz;   // gated activations of block b
  float* gx_out;         // g_x(b+1) [rows][R]
  float* gu_out;         // g_u(b) [rows][2D]
  const float* wx16;     // fp16 hi|lo image A[R][KS*2D] of block b+1's reversed conv
  const float* wu16;     // fp16 hi|lo image A[D][R + F0] = [W_r(b) | V(b)]
  const float* am_gu_in; const float* am_gf;    // running max-abs of g_u(b+1) / dL/da
  float* am_gx; float* am_gu;                   // running max-abs of the two outputs
  int32_t B, T, dil;     // dil = dilation of block b+1
};
int wn_bwd_pair_supported(int R, int D, int KS, int F0);
int wn_launch_bwd_pair(const WnBwdPairArgs& a, hipStream_t s);
// the same two products for 128-channel blocks, weight images streamed through an LDS ring (wn_bwd16s.hip)
int wn_bwd_s128_supported(int R, int D, int KS, int F0);
int wn_launch_bwd_s128(const WnBwdPairArgs& a, hipStream_t s);
// dW_s / db_s of the folded skip path for all blocks (wn_wgrad_skip.hip)
int wn_wgrad_skip_supported(int D, int S, int KZ);
int wn_launch_wgrad_skip(const float* z, int ldz, const float* g, int ldg, int64_t rows, int KZ, int S, int D,
                         int nsplit, float* slab, int64_t P, int64_t w_off0, int64_t w_stride, int64_t b_off0,
                         int64_t b_stride, int nblocks, const float* gmax, hipStream_t s);
int wn_wgrad_tile_k();
int wn_wgrad_tile_n();
int wn_launch_wgrad_batched(const WnWgJob* d_jobs, int njobs, float* ws, float* slab, int64_t P, int B, int T,
                            int splits_per_b, hipStream_t s, bool exact_fp32 = false);
int wn_launch_reduce_table(const float* slab, int nsplit, int64_t P, float* out, const WnTensorDesc* d_table,
                           int n, hipStream_t s, const WnTensorDesc* h_table = nullptr);
// out[(k / seg_len) * seg_stride + (k % seg_len) * N + n] (+)= sum_s slab[s][k][n]
struct WnReduceArgs {
  const float* slab; int32_t nsplit; int32_t K; int32_t N;
  float* out; int32_t seg_len; int64_t seg_stride; int32_t accumulate;
  int32_t replicate; int64_t rep_stride;   // write the same result to `replicate` places
};
int wn_launch_reduce(const WnReduceArgs& a, hipStream_t s);

// ---------------------------------------------------------------- fused residual block forward
struct WnLayerFwdArgs {
  const float* x;        // [B*T][R]
  const float* frag_d;   // KS images (tap-major) of A[2D][R]
  const float* frag_r;   // image of A[R][D]
  const float* bias_d;   // [2D]
  const float* bias_r;   // [R]
  const float* cb;       // [B][2D] conditioning bias or null
  float* x_out;          // [B*T][R]
  float* o_out;          // [B*T][R] pre-residual output (skip when skip_channels is None) or null
  float* z_out; int32_t ldz;   // gated activations, row stride ldz, or null
  float* ag_out;         // [B*T][D] saved sigmoid (tanh is recovered as z / sigmoid in backward), or null
  const float* res;      // residual source [B*T][R] when it is not the conv input (depth > 1), or null
  const float* xt[3];    // queued generation: tap j reads rows of xt[j] (no time shift) instead of x; or null
  int32_t B, T, R, D, KS, dilation, residual;
  float* absmax_out;     // split-precision kernel: running max-abs of x_out (forward range guard), or null
};
int wn_layer_fwd_supported(int R, int D, int KS);
// fp16 hi/lo split (3-product) variant with LDS-resident weights; frag_d / frag_r are kind-1 images
int wn_layer_fwd_f16_supported(int R, int D, int KS);
int wn_launch_layer_fwd_f16(const WnLayerFwdArgs& a, hipStream_t s);
size_t wn_frag16_floats(int I, int KK);
// streamed-weights form for blocks too wide for LDS-resident images (R = D = 128; wn_layer16s.hip): frag_d = kind-1 image
// A[2D][KS*R] in natural row-tile order, frag_r = kind-1 image A[R][D]
int wn_layer_fwd_s128_supported(int R, int D, int KS);
int wn_launch_layer_fwd_s128(const WnLayerFwdArgs& a, hipStream_t s);
int wn_launch_layer_fwd(const WnLayerFwdArgs& a, hipStream_t s);
// one queued-generation step of a 128-channel block, rows = utterances (wn_gen128.hip): same arguments as the streamed
// forward with T == 1 and both taps given as xt[0], xt[1]
int wn_gen_block128_supported(int R, int D, int KS);
int wn_launch_gen_block128(const WnLayerFwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------- fused generation step (wn_gen.hip)
struct WnGenBlock {
  int64_t ring_off;      // workspace offset of this block's ring [nslots][B][R]
  int64_t w16d_off;      // workspace offset of the fp16 split image of the gated conv
  int64_t w16r_off;      // ... of conv1
  int64_t bias_d_off;    // parameter offsets
  int64_t bias_r_off;
  int64_t cb_off;        // workspace offset of the [B][2D] conditioning bias, or < 0
  int32_t nslots, dilation;
};
struct WnGenStepArgs {
  const float* params;
  float* ws;                       // workspace base (rings, images, row buffers)
  const WnGenBlock* blocks;        // device table
  const float* xin;                // [KS][B] raw sample ring
  const float* causal_w;           // (KS, 1, R)
  const float* causal_b;
  int64_t zrow_off;                // [N][B][D] gated activations of this step
  int64_t hrow_off;                // [B][R] last block output (use_skip False) or < 0
  int64_t skip_w16_off;            // fp16 split image of the folded skip contraction (skip_tiles > 0)
  int64_t skip_bias_off;           // summed skip biases
  int64_t skiprow_off;             // [B][skip_ld] folded skip sum of this step
  int32_t skip_tiles, skip_ld;     // skip waves (column tiles of 32) carried by the chain kernel, or 0
  int32_t skip_act, pad1_;         // activation of the folded skip sum (the head's first conv when the plan folds it in)
  int64_t u0_off;                  // [N][tiles][2D/32 * 1024] partial gated-conv accumulators (pre kernel -> chain)
  int64_t tau;
  int32_t B, nblocks, residual;
  int32_t ntiles;                  // utterance tiles = chain workgroups (the chain launch appends helper workgroups)
  unsigned long long* ts;          // phase stamps (debug switch 24) or null
  int64_t bias_r_off0, bias_r_stride;   // conv1 bias of block b at params + off0 + b * stride (stride 0: not uniform, read the table)
  WnGenBlock blk0[3];              // blocks[0..2] by value (the first fetches do not wait for the table)
  float* guard;                    // range guard of the generate call: running max-abs of the residual stream and of the
                                   // folded skip sum (the split-precision casts need |.| < 65504), or null
};
// every block of one queued-generation step for 128-channel blocks in one launch (wn_gen128.hip); the table's w16d_off is
// the natural-order image of the gated conv (BlockInfo::f16nat)
struct WnGen128Args {
  const float* params;
  float* ws;
  const WnGenBlock* blocks;        // device table
  int64_t zrow_off;                // [N][B][D] gated activations of this step
  int64_t hrow_off;                // [B][R] last block output (use_skip False) or < 0
  int64_t tau;
  int32_t B, nblocks, residual;
  float* guard;                    // range guard of the generate call or null
  // folded skip contraction with 128 columns inside the chain (waves 4..7), or skip_w16_off < 0
  int64_t skip_w16_off, skip_bias_off, skiprow_off;
  int32_t skip_act;
  // input causal conv (kernel size 2, one input channel) inside the chain: raw sample ring [2][B], kernel (2, 1, R), bias;
  // or xin null: block 0's ring slot was written by an earlier launch
  const float* xin; const float* causal_w; const float* causal_b;
  // relay form (wn_gen_relay128_kernel): utterance tiles, workspace offset of the granule areas, the step's epoch (>= 1),
  // the word a reader that gave up waiting writes
  int32_t ntiles; uint32_t epoch; int64_t relay_off; unsigned* tmo;
  unsigned long long* ts;          // phase stamps (knob 24) or null
  int32_t mute_block;              // fault injection (knob 3): this block withholds its hand-over; -1 = none
};
int wn_launch_gen_chain128(const WnGen128Args& a, hipStream_t s);
int wn_launch_gen_relay128(const WnGen128Args& a, hipStream_t s);
int64_t wn_gen_relay128_floats(int B, int nblocks);
// queued generation: where a sampler also puts its sample (output rows [rows][length] at column step; network input slot)
struct WnEmit { float* out; int length; int step; float* xin_slot; };
// the head of a generation step in one launch (wn_gen.hip)
#define WN_GEN_HEAD_MAX 4
struct WnGenHeadArgs {
  const float* params;
  float* ws;
  int64_t in_off;                  // [B][in_ld] input rows (folded skip sum or last block output)
  int64_t out_off;                 // [B][N[last]] logits
  int64_t w16_off[WN_GEN_HEAD_MAX];   // fp16 split images A[N][K] (workspace offsets)
  int64_t bias_off[WN_GEN_HEAD_MAX];  // parameter offsets
  int32_t K[WN_GEN_HEAD_MAX], N[WN_GEN_HEAD_MAX], act[WN_GEN_HEAD_MAX];
  int32_t in_ld, nlayers, B;
  // an exact-fp32 last layer behind the split-precision ones (f32_K > 0): the rows of wn_gemm_rows_kernel<1> -- at most 32
  // columns, e.g. the 3 x mixtures outputs of a mixture head, which no split-precision image covers
  int64_t f32_w_off, f32_bias_off; // fp32 fragment image (workspace offset), bias (parameter offset)
  int32_t f32_K, f32_N;
  // sampling tail in the same launch: 0 none, 1 categorical deterministic (arg max), 2 categorical stochastic draw,
  // 3 mixture deterministic, 4 mixture stochastic (mix_M components, mix_kind 1 logistic / 2 gaussian)
  int32_t tail, mix_M, mix_kind;
  float inv_lv;                    // 2 / levels
  uint64_t seed, offset;           // Philox key / counter word of a stochastic draw
  float* samp;                     // [B] samples (or null)
  WnEmit em;                       // output rows / network input slot
  float* guard;                    // range guard of the generate call (hidden activations are cast to fp16 hi | lo), or null
};
int wn_launch_gen_head(const WnGenHeadArgs& a, hipStream_t s);
int wn_gen_blocks_supported(int R, int D, int KS);
int wn_gen_chain_max_blocks();
int64_t wn_gen_u0_floats(int B, int nblocks, int D);
int wn_gen_skip_fusable(int S);
int wn_launch_gen_blocks(const WnGenStepArgs& a, int R, int KS, int what, hipStream_t s);
int wn_launch_gen_head_pre(const WnGenHeadArgs& a, const WnGenStepArgs& g, int R, int KS, hipStream_t s);

// ---------------------------------------------------------------- elementwise / loss / sampling
int wn_launch_add(const float* a, const float* b, float* out, int64_t n, hipStream_t s);
int wn_launch_fill(float* p, float v, int64_t n, hipStream_t s);
int wn_launch_dact_mul(const float* g, const float* y, float* out, int64_t n, int act, hipStream_t s);
uint32_t wn_dropout_key(uint64_t seed, int block, uint64_t step);
int wn_launch_dropout(const float* x, const float* g_res, float* out, int64_t n, float rate, uint32_t key,
                      float* absmax_out, hipStream_t s);
int wn_launch_gate(const float* u, int64_t rows, int D, float* ag, float* z, int ldz, hipStream_t s);
// global conditioning of all blocks at once (wn_elem.hip)
int wn_launch_cond_scatter(const float* tmp, const float* params, int64_t b_off0, int64_t b_stride, int B, int N, int D2,
                           float* cb, hipStream_t s);
int wn_launch_cond_gather(const float* slab, int64_t P, int spb, int64_t bd_off0, int64_t bd_stride, int B, int N, int D2,
                          float* dcb, hipStream_t s);
int wn_launch_cond_wgrad(const float* m, const float* dcb, int B, int Cc, int N, int D2, float* grads, int64_t w_off0,
                         int64_t w_stride, int64_t b_off0, int64_t b_stride, hipStream_t s);
// skip path folded into the head's first convolution (wn_elem.hip): V(b) = W_s(b) W_f0, b' = b_f0 + W_f0^T sum b_s;
// and the weight gradients of conv_skip (all blocks) and of that convolution from M = Z^T dL/da
int wn_launch_skip_fold(const float* params, int64_t ws_off0, int64_t ws_stride, int64_t wf0_off, int64_t bf0_off,
                        const float* bsum, int N, int D, int S, int F0, float* V, float* bfold, float* wsall, hipStream_t s);
// C[i][j] = sum_k A[i * sai + k * sak] * B[k * sbk + j * sbj] (small fp32 product, LDS-tiled)
int wn_launch_sgemm_small(const float* A, int64_t sai, int64_t sak, const float* B, int64_t sbk, int64_t sbj, float* C, int ldc,
                          int M, int N, int K, hipStream_t s);
int wn_launch_sgemm_small_batched(const float* A, int64_t sai, int64_t sak, int64_t za, const float* B, int64_t sbk, int64_t sbj,
                                  int64_t zb, float* C, int ldc, int64_t zc, int M, int N, int K, int nz, const float* bias,
                                  int act, hipStream_t s, int zk = 0);
int wn_launch_skip_scatter(const float* Y, const float* colsum, int64_t ws_off0, int64_t ws_stride, int64_t bs_off0,
                           int64_t bs_stride, int64_t bf0_off, int N, int D, int S, int F0, float* grads, hipStream_t s);
// dW, db of the input causal conv into the batched weight-gradient slab (wn_elem.hip)
int wn_inconv_wgrad_supported(int R, int KS);
int wn_launch_inconv_wgrad(const float* x, const float* g, int B, int T, int R, int KS, int splits_per_b, float* slab,
                           int64_t P, int64_t w_off, int64_t b_off, hipStream_t s);
int wn_launch_inconv_fwd(const float* x, const float* w, const float* bias, int B, int T, int R, int KS, float* y,
                         float* absmax_out, hipStream_t s);
// out[0] = 1 when the split-precision kernels were fed a forward activation at or beyond `limit` (or a non-finite one)
int wn_launch_guard_flag(const float* absmax, float limit, int enabled, float* out, hipStream_t s);
// dst[0] = max(dst[0], src[0]) on the bit patterns (non-negative floats; inf / NaN stay on top)
int wn_launch_guard_accumulate(const float* src, float* dst, hipStream_t s);
int wn_launch_gen_tail_cat_det(const float* logits, int rows, int C, int bits, float* out, int length, int step,
                               float* xin_slot, hipStream_t s);
int wn_launch_batch_reduce(const float* slab, int B, int splits, int N, float* out, hipStream_t s);
int64_t wn_colsum_scratch_floats(int B, int C);
int wn_launch_colsum_per_batch(const float* g, int B, int T, int C, float* out, float* scratch, hipStream_t s);
int wn_launch_quantize(const float* x, int32_t* idx, int64_t n, int bits, hipStream_t s);
int wn_launch_dequantize(const int32_t* idx, float* x, int64_t n, int bits, hipStream_t s);
int wn_launch_mulaw(const float* x, float* y, int64_t n, hipStream_t s);
int wn_launch_inv_mulaw(const float* y, float* x, int64_t n, hipStream_t s);
int wn_launch_softmax(const float* logits, float* probs, int64_t rows, int C, hipStream_t s);
int wn_sample_from_logits_supported(int C);
int wn_launch_sample_rand_cat_logits(const float* logits, int64_t rows, int C, int bits, uint64_t seed, uint64_t offset,
                                     float* out, hipStream_t s);
// categorical: Keras sparse CE on clipped probabilities; g_logits may be null (loss only)
int wn_launch_cat_loss(const float* logits, const int32_t* target, int64_t rows, int C,
                       float gscale, float* loss_rows, float* g_logits, float* absmax_out, hipStream_t s,
                       float* sample_out = nullptr, int bits = 8, uint64_t seed = 0, uint64_t offset = 0);
// from_probs variant used by WaveNet.loss_fn(target, pred) on materialised probabilities
int wn_launch_cat_loss_probs(const float* probs, const int32_t* target, int64_t rows, int C,
                             float* loss_rows, hipStream_t s);
// kind 1 = logistic, 2 = gaussian
int wn_launch_mix_loss(const float* pred, const float* y, int64_t rows, int M, int bits, int kind,
                       float gscale, float* loss_rows, float* g_pred, float* absmax_out, hipStream_t s);
int wn_launch_sum(const float* v, int64_t n, float scale, float* out, float* scratch, hipStream_t s);
int wn_launch_sqdiff_sum(const float* a, const float* b, int64_t n, float scale, float* out, float* scratch, hipStream_t s);
// deterministic samplers: categorical argmax -> left bin edge; mixtures -> clipped mean
int wn_launch_sample_det_emit(const float* pred, int64_t rows, int C, int M, int bits, float* out, WnEmit em, hipStream_t s);
int wn_launch_sample_rand_emit(const float* pred, int64_t rows, int C, int M, int bits, int kind, uint64_t seed, uint64_t offset,
                               float* out, WnEmit em, hipStream_t s);
int wn_launch_sample_rand_cat_logits_emit(const float* logits, int64_t rows, int C, int bits, uint64_t seed, uint64_t offset,
                                          float* out, WnEmit em, hipStream_t s);
int wn_launch_sample_det(const float* pred, int64_t rows, int C, int M, int bits, float* out,
                         hipStream_t s);
// stochastic samplers (Philox4x32-10 keyed by seed, counter = row)
int wn_launch_sample_rand(const float* pred, int64_t rows, int C, int M, int bits, int kind,
                          uint64_t seed, uint64_t offset, float* out, hipStream_t s);

// ---------------------------------------------------------------- optimizer
int wn_launch_sumsq(const float* g, const WnTensorDesc* d_table, int n, float* norms2, hipStream_t s);
int wn_launch_axpy_table(float* y, const float* x, const WnTensorDesc* d_table, int n, float coef,
                         hipStream_t s);
int wn_launch_adam(float* p, const float* g, float* m, float* v, const WnTensorDesc* d_table, int n,
                   const float* norms2, float clipnorm, float alpha, float beta1, float beta2,
                   float eps, const float* skip_flag, hipStream_t s);
