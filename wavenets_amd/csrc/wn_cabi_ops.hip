// ==========================================================================================
// elementwise entry points
// ==========================================================================================
#include "wn_plan_internal.h"

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
  wnp::ex(p).step_sample = sample_out; wnp::ex(p).step_sample_det = deterministic; wnp::ex(p).step_sample_seed = seed; wnp::ex(p).step_sample_off = offset;
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

