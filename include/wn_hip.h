/* libwn_hip.so -- C-ABI of the MI355X (gfx950) WaveNet training / generation hot path.
 *
 * Drop-in boundary for jirsat/wavenets (reference paths relative to /root/reference).  The
 * reference has no FFI: its boundary is the Keras class surface of src/layers.py
 * (WaveNetLayer) and src/model.py (WaveNet).  The Python mirror of those classes
 * (wavenets_amd/layers.py, wavenets_amd/model.py) binds exactly the entry points declared
 * here through ctypes; each one cites the reference method it replaces.
 *
 * Conventions
 *  - all tensors are device pointers to contiguous fp32 (or int32 where stated) buffers in the
 *    reference's channels-last layout (B, T, C); Conv1D kernels (k, C_in, C_out), Dense
 *    kernels (in, out), biases (C_out)                        [src/layers.py:134]
 *  - parameters / gradients / Adam moments are ONE flat fp32 buffer each, tensors in Keras
 *    creation order (causal; per block: dilated stack, conv1, conv_skip, conv_cond; final
 *    convs; mapping Dense stack) -- wn_plan_tensor_info() enumerates them
 *  - no function allocates or frees caller-visible memory: outputs, saved activations and
 *    scratch live in a caller-provided workspace of wn_plan_workspace_floats() floats
 *  - every launch is asynchronous on the given hipStream_t (passed as void*); no host sync
 *  - return value: 0 = ok, <0 = WN_E_* ; message via wn_last_error_string() (thread-local)
 */
#ifndef WN_HIP_H
#define WN_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_OK 0
#define WN_E_INVALID (-1)      /* bad argument (the Python layer raises ValueError) */
#define WN_E_UNSUPPORTED (-2)  /* shape / option outside what the kernels cover */
#define WN_E_HIP (-3)          /* HIP runtime error */

/* activation ids (Keras activation strings used by the reference configs, train.py:35,38) */
#define WN_ACT_LINEAR 0
#define WN_ACT_RELU 1
#define WN_ACT_LEAKY_RELU 2
#define WN_ACT_TANH 3
#define WN_ACT_SIGMOID 4
#define WN_ACT_ELU 5

/* output head / sampling_function (src/model.py:65-69) */
#define WN_HEAD_CATEGORICAL 0
#define WN_HEAD_LOGISTIC 1
#define WN_HEAD_GAUSSIAN 2

#define WN_MAX_FINAL 8
#define WN_MAX_MAPPING 8

/* Mirror of the WaveNet constructor keywords, src/model.py:14-34 */
typedef struct wn_config {
  int32_t kernel_size;
  int32_t channels;
  int32_t blocks;
  int32_t layers_per_block;
  int32_t activation;           /* WN_ACT_* for non-gated convs and the head */
  int32_t dilation_bound;
  int32_t num_mixtures;         /* 0 = None */
  int32_t head;                 /* WN_HEAD_* */
  int32_t bits;
  int32_t skip_channels;        /* 0 = None */
  int32_t dilation_channels;    /* 0 = None (-> channels, src/layers.py:49-50) */
  int32_t use_residual;
  int32_t use_skip;
  int32_t n_final;              /* len(final_layers_channels) */
  int32_t final_channels[WN_MAX_FINAL];
  int32_t cond_inputs;          /* 0 = conditioning None; else width of the raw global condition */
  int32_t n_mapping;            /* len(mapping_layers) */
  int32_t mapping_channels[WN_MAX_MAPPING];
  int32_t mapping_activation;
  float l2_reg_factor;
} wn_config;

typedef struct wn_plan wn_plan;

const char* wn_last_error_string(void);

/* ---- plan (WaveNet.__init__ + build, src/model.py:14-211).  The network description inside a plan (shapes, dilation
 * schedule, tensor table, image layouts, workspace layouts) is immutable after wn_plan_create; device copies of its small
 * tables are made once, on first use, under a lock.  The per-caller launch state the calls below set and the entry
 * points read -- dropout counter (wn_plan_set_dropout), armed step-sample pointer (wn_plan_arm_step_sample), phase
 * selection (wn_plan_set_train_phases), side stream / events, measurement hooks (wn_prof_*, wn_stack_prof_*, wn_phase_*),
 * device-side job tables cached per (B, T) -- is NOT in the plan but in an execution state (wn_exec, below): the
 * plan's own by default, or the one the calling thread has bound.  The setters named wn_plan_* act on that state. ---- */
wn_plan* wn_plan_create(const wn_config* cfg);          /* NULL on error */
void wn_plan_destroy(wn_plan* p);

/* Execution state.  A wn_plan is immutable once created (configuration, tensor table, weight-image layouts, workspace
 * layouts) and may be shared by any number of host threads and streams.  What changes while a plan is
 * USED -- the dropout counter, the armed step sample, the phase selection, the weight-gradient job tables and generation
 * block table cached for the last (B, T) (device memory), the side stream and its events, the profiling events -- lives
 * in a wn_exec.  Every plan carries one of its own, used by callers that never bind another: the single-threaded use of
 * the reference (one Python thread drives everything, train.py:203-258) needs nothing below.  A caller that drives ONE
 * plan from several threads / streams creates one wn_exec per thread and binds it there; the binding is thread-local
 * and applies to calls on the plan the state was created for.  A new state copies the dropout setting and phase
 * selection the plan's own state has at that moment.  wn_exec_bind(NULL) unbinds. */
typedef struct wn_exec wn_exec;
wn_exec* wn_exec_create(const wn_plan* p);
void wn_exec_destroy(wn_exec* e);
int wn_exec_bind(wn_exec* e);
int64_t wn_plan_param_count(const wn_plan* p);
int32_t wn_plan_num_tensors(const wn_plan* p);
/* tensor idx in Keras creation order: flat offset, element count, rank, shape[3] */
int wn_plan_tensor_info(const wn_plan* p, int32_t idx, int64_t* offset, int64_t* len,
                        int32_t* ndim, int64_t* shape3, int32_t* is_kernel);
int32_t wn_plan_receptive_field(const wn_plan* p);      /* src/model.py:122 */
int32_t wn_plan_out_channels(const wn_plan* p);
int32_t wn_plan_dilation(const wn_plan* p, int32_t conv_index);   /* src/model.py:79-81 */
/* workspace size in floats for a (B, T) call; training != 0 keeps activations for backward */
int64_t wn_plan_workspace_floats(const wn_plan* p, int32_t B, int32_t T, int32_t training);

/* ---- Dropout(rate) on every block input in training calls (src/layers.py:108-111,195-196; WaveNet
 * keyword `dropout`).  The keep-mask is a counter-based hash of (seed, block, step, element); `step`
 * is used as given by every following training call: the CALLER owns the counter (the Python mirror sets
 * step = call_index * world_size + rank + 1 before each step, so that replicas draw independent masks as
 * under MirroredStrategy, and stores call_index in its checkpoints).  rate 0 disables. */
int wn_plan_set_dropout(wn_plan* p, float rate, uint64_t seed, uint64_t step);
uint32_t wn_dropout_key_for(uint64_t seed, int32_t block, uint64_t step);   /* test hook */

/* ---- measurement hook (bench.py): HIP events on the caller's stream around the residual-block
 * forward launches -- one pair around the whole chain when it is N back-to-back launches of the fused
 * block kernel (a pair per launch would time its own event packets too), else one pair per block;
 * wn_prof_read returns launches and the average per-launch time after a stream sync.
 * Not part of the reference surface. */
int wn_debug_set(int key, int value);     /* kernel-variant switches of the CALLING THREAD (list: csrc/wn_error.cpp);
                                             key 1 = 1 selects the exact-fp32 MFMA kernels */
int wn_debug_value(int key);              /* current value of a switch in the calling thread */
int wn_debug_gen_ts(unsigned long long* out_96); /* switch 24: s_memtime stamps [role 3][block 4][phase 8] of the last
                                             generation chain launch (profiling hook; -1 if none was taken) */
/* one line naming the kernel family every phase of a pass selects for this plan under the calling thread's switches */
int wn_plan_describe(const wn_plan* p, char* buf, int32_t len);
int wn_prof_enable(wn_plan* p, int32_t max_launches);
int wn_prof_read(wn_plan* p, int32_t* launches, float* avg_ms);
/* test / diagnosis hook: float offset and length, inside the caller's TRAINING workspace for (B, T), of an
 * intermediate the last wn_train_fwd_bwd left there (deferred weight-gradient layout, layers_per_block = 1).
 * what: 0 H[idx] | 1 Z[idx] | 2 saved sigmoid[idx] | 3 skip sum | 4 head activation[idx] | 5 logits |
 *       6 dL/d(final[idx] pre-activation) | 7 dL/d(skip sum) | 8 dL/du[idx] | 9 dL/dH[idx] | 10 max-abs slots |
 *       11 activated output of non-gated dilated conv i of block b, idx = b * (layers_per_block - 1) + i (any layout) */
int wn_debug_ws_region(const wn_plan* p, int32_t B, int32_t T, int32_t what, int32_t idx, int64_t* off,
                       int64_t* len);
/* the whole residual-block stack of a forward pass: one event pair per pass from the first block launch to
 * the end of the folded skip contraction = t_stack_fwd of SURVEY.md 8(d); read returns passes and the average */
int wn_stack_prof_enable(wn_plan* p, int32_t max_passes);
int wn_stack_prof_read(wn_plan* p, int32_t* passes, float* avg_ms);
/* the per-pass weight-space preparation of the folded skip path of the same passes (bias sum, V = W_s W_f0, fp16 images;
 * it runs before the first block launch, outside the pair above): average per pass, 0 when the plan does not fold */
int wn_stack_prof_read_foldprep(wn_plan* p, int32_t* passes, float* avg_ms);
/* phase marks of wn_train_fwd_bwd: ms4 = {forward, loss, backward-data chain, weight gradients + rest}
 * of the last call (read after a stream sync) */
int wn_phase_enable(wn_plan* p, int32_t on);
int wn_phase_read(wn_plan* p, float* ms4);

/* ---- WaveNet.call, src/model.py:213-239 ----
 * x (B,T,1); cond (B, cond_inputs) or NULL; out (B,T,C_out): probabilities (categorical) or
 * linear mixture parameters; logits_out optional (B,T,C_out) pre-softmax. */
int wn_forward(wn_plan* p, const float* params, const float* x, const float* cond, int32_t B,
               int32_t T, float* out, float* logits_out, float* workspace, int64_t ws_floats,
               void* stream);

/* WaveNet.call(inputs, training=True): the same forward with the Dropout layers active (src/layers.py:195-196),
 * mask of the step last set by wn_plan_set_dropout; workspace sized for training (wn_plan_workspace_floats(.., 1)) */
int wn_forward_training(wn_plan* p, const float* params, const float* x, const float* cond, int32_t B,
                        int32_t T, float* out, float* logits_out, float* workspace, int64_t ws_floats,
                        void* stream);

/* ---- gradient half of WaveNet.train_step, src/model.py:319-335 ----
 * x_full (B,T+1,1): inputs = x[:, :-1], targets = prepare_target(x[:, 1:]).
 * loss = sum_{b,t} l / global_batch (+ l2 * sum W^2 / n_replicas).  grads receives
 * d(loss)/d(params) for THIS replica's rows (to be SUM-all-reduced by the caller).
 * loss_out: 3 device floats {loss, reg_loss, range_flag}.  pred_out optional (B,T,C_out). */
int wn_train_fwd_bwd(wn_plan* p, const float* params, const float* x_full, const float* cond,
                     int32_t B, int32_t T, int32_t global_batch, int32_t n_replicas, float* grads,
                     float* loss_out, float* pred_out, float* workspace, int64_t ws_floats,
                     void* stream);

/* ---- WaveNet.test_step loss, src/model.py:362-381 (forward + loss only); loss_out: 3 device floats as above ---- */
int wn_eval_loss(wn_plan* p, const float* params, const float* x_full, const float* cond,
                 int32_t B, int32_t T, int32_t global_batch, float* loss_out, float* pred_out,
                 float* workspace, int64_t ws_floats, void* stream);

/* ---- optimizer.apply_gradients with Adam(lr, clipnorm), train.py:225-226, model.py:336 ----
 * Keras Adam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); p -= alpha*m/(sqrt(v)+eps); per-tensor
 * tf.clip_by_norm when clipnorm > 0.  step is 1-based.  scratch: >= num_tensors floats. */
int wn_adam_step(wn_plan* p, float* params, const float* grads, float* m, float* v, int64_t step,
                 float lr, float beta1, float beta2, float eps, float clipnorm, float* scratch,
                 void* stream);
/* the same update, skipped on the device when *skip_flag != 0 (the range flag of the step, after the data-parallel
 * SUM: see "forward range guard" below); skip_flag may be NULL */
int wn_adam_step_guarded(wn_plan* p, float* params, const float* grads, float* m, float* v, int64_t step,
                         float lr, float beta1, float beta2, float eps, float clipnorm, float* scratch,
                         const float* skip_flag, void* stream);

/* ---- forward range guard of the split-precision mode ----
 * The default kernels evaluate fp32 products from fp16 hi|lo operand splits; an activation beyond the fp16 range
 * (65504) would turn into inf/NaN.  Every forward pass therefore keeps the running max-abs of the tensors that feed
 * such kernels (residual stream, skip sum, head activations) in one workspace float: wn_train_fwd_bwd / wn_eval_loss
 * write loss_out[2] = 1 when it reached wn_range_limit() (else 0; always 0 in exact-fp32 mode); wn_forward callers
 * read the float at workspace[wn_plan_range_slot()].  A tripped pass must be repeated with the exact-fp32 kernels
 * (wn_debug_set(1, 1)): the Python mirror does (WaveNet.train_step / call / test_step). */
int64_t wn_plan_range_slot(const wn_plan* p, int32_t B, int32_t T, int32_t training);
float wn_range_limit(void);

/* ---- WaveNet.generate / _generation, src/model.py:241-307 (intended semantics) ----
 * window (B,RF,1) initial samples; out (B,length,1).  deterministic != 0: argmax / mode
 * sampling; else Philox draws keyed by seed.  queued != 0 uses per-layer ring buffers
 * (result-identical to the sliding window; the reference's README.md:16 TODO).
 * A queued step of 128-channel blocks is ONE launch of one workgroup per (block, 32 utterances) that hand their rows
 * on inside the launch (wn_gen_relay128_kernel); it is taken while blocks x ceil(B / 32) fits the device's CU count,
 * every wait in it is bounded, and a wait that gave up is reported through the word behind wn_generate_guard_slot. */
int wn_generate(wn_plan* p, const float* params, const float* window, const float* cond, int32_t B,
                int32_t length, int32_t deterministic, int32_t queued, uint64_t seed, float* out,
                float* workspace, int64_t ws_floats, void* stream);
int64_t wn_generate_workspace_floats(const wn_plan* p, int32_t B, int32_t queued);
/* float offset, inside the caller's generation workspace, of the call's range-guard slot.  After wn_generate (stream
 * synchronised) the float is >= wn_range_limit() if and only if some split-precision kernel of the call was fed an
 * |activation| at or beyond that limit: the priming pass (residual stream, skip sum, head activations), and in EVERY
 * queued step the residual stream and folded skip sum of the chain kernels, the per-block kernels, the skip contraction
 * and the hidden head activations.  (Below the limit it holds the priming pass's largest magnitude; the per-step kernels
 * only ever raise it past the limit.)  A tripped call is to be repeated with the exact-fp32 kernels (wn_debug_set(1, 1)).
 * The 32-bit word right behind it (slot + 1) is the hand-off watchdog of the multi-workgroup generation step
 * (wn_gen_relay128_kernel): zero after a good call; non-zero bits name a workgroup that gave up waiting for its
 * predecessor's rows -- the samples of such a call are invalid and the caller must treat it as failed. */
int64_t wn_generate_guard_slot(const wn_plan* p, int32_t B, int32_t queued);

/* ---- WaveNetLayer.call, src/layers.py:178-224, standalone block ----
 * weights in Keras layout: dil_kernels = layers_in_block kernels concatenated
 * [(k,Cin_i,Cout_i)...], dil_biases likewise; conv1 (D,R); conv_skip (D,S) or NULL;
 * conv_cond (Cc,2D) or NULL with cond (B,T,Cc).  dilations[depth].
 * saved: caller buffer of wn_layer_saved_floats() floats kept for wn_layer_bwd (or NULL). */
typedef struct wn_layer_desc {
  int32_t kernel_size, channels, dilation_channels, skip_channels /*0 = None*/;
  int32_t depth;                 /* len(dilation_rate) */
  int32_t dilations[16];
  int32_t activation, residual, cond_channels /*0 = no condition*/;
  int32_t in_channels;           /* channels of the input tensor */
} wn_layer_desc;
int64_t wn_layer_saved_floats(const wn_layer_desc* d, int32_t B, int32_t T);
int64_t wn_layer_workspace_floats(const wn_layer_desc* d, int32_t B, int32_t T);
int wn_layer_fwd(const wn_layer_desc* d, const float* params, const float* x, const float* cond,
                 int32_t B, int32_t T, float* x_out, float* skip_out, float* saved,
                 float* workspace, void* stream);
int wn_layer_bwd(const wn_layer_desc* d, const float* params, const float* x, const float* cond,
                 const float* saved, const float* g_x_out, const float* g_skip, int32_t B, int32_t T,
                 float* g_x, float* g_cond, float* g_params, float* workspace, void* stream);
int64_t wn_layer_param_count(const wn_layer_desc* d);

/* ---- elementwise pieces of the boundary ---- */
/* prepare_target = Discretization (tf Bucketize), src/model.py:151-153: bit-exact indices */
int wn_quantize(const float* x, int32_t* idx, int64_t n, int32_t bits, void* stream);
/* index -> left bin edge, src/model.py:411,418 */
int wn_dequantize(const int32_t* idx, float* x, int64_t n, int32_t bits, void* stream);
/* mu-law companding src/utils.py:34-35 and its inverse src/callbacks.py:126-131 */
int wn_mulaw(const float* x, float* y, int64_t n, void* stream);
int wn_inv_mulaw(const float* y, float* x, int64_t n, void* stream);
/* WaveNet.loss_fn(target, pred), src/model.py:505-551: per-(b,t) loss, rows = B*T.
 * categorical: target int32 indices, pred probabilities; mixtures: target fp32 values. */
int wn_loss_fn(int32_t head, const void* target, const float* pred, int64_t rows, int32_t C,
               int32_t num_mixtures, int32_t bits, float* loss_rows, void* stream);
/* Arms the NEXT wn_train_fwd_bwd on this plan to also write sample = sample_waveform(pred) of that step
 * (rows = B*T floats; src/model.py:338, consumed by the compiled metrics :340-346) drawn from the logits inside
 * the step -- the same draw wn_sample_waveform(seed, offset) makes from that step's pred, without the (rows, C)
 * probability tensor.  sample_out = NULL disarms.  WN_E_UNSUPPORTED (nothing armed): categorical head with a
 * deterministic draw or more than 1024 classes; use pred_out + wn_sample_waveform there. */
int wn_plan_arm_step_sample(wn_plan* plan, float* sample_out, int32_t deterministic, uint64_t seed, uint64_t offset);
/* wn_train_fwd_bwd in two calls: phases = 1 runs forward + loss (+ the armed step sample) and returns, phases = 2 runs the
 * backward pass and the weight gradients of THAT forward pass (same arguments, same workspace), 3 = both (default).  A
 * host driver uses the gap to queue its read-back of loss / metrics 4 ms before the step ends. */
int wn_plan_set_train_phases(wn_plan* plan, int32_t phases);
/* tf.keras.metrics.MeanSquaredError(y_true, sample) of one step (train.py:227, src/model.py:338-346) as a device scalar:
 * out[0] = scale * sum_i (a[i] - b[i])^2 in double accumulation; scale = 1 / (n * replicas) makes the SUM over the
 * replicas the metric's mean.  scratch: >= 2048 floats of the caller's. */
int wn_sum_squared_error(const float* a, const float* b, int64_t n, float scale, float* out, float* scratch,
                         void* stream);
/* WaveNet.sample_waveform(pred, deterministic), src/model.py:393-503: (rows,C) -> (rows) */
int wn_sample_waveform(int32_t head, const float* pred, int64_t rows, int32_t C,
                       int32_t num_mixtures, int32_t bits, int32_t deterministic, uint64_t seed,
                       uint64_t offset, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WN_HIP_H */
