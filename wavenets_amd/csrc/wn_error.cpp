// thread-local error string of the C-ABI (wn_last_error_string)
#include <cstdarg>
#include <cstdio>
static thread_local char g_wn_err[512] = "";
void wn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_wn_err, sizeof(g_wn_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* wn_last_error_string(void) { return g_wn_err; }

// debug / tuning knobs (not part of the reference surface): small integer registers read by the
// launchers, settable from tools/ scripts without rebuilding
//    0  fused fp32 block-forward kernel variant / ablations
//    0  = -1: the head's weight gradients share the blocks' slab and time-split count; > 0: that many time splits per utterance
//    1  = 1: exact-fp32 MFMA kernels everywhere (no fp16 hi/lo split)
//    2  = 1: no resident-weights rows GEMM (streamed form instead)
//    3  = 1: weight gradients alone stay exact fp32
//    4  = 1: precompute W_s g_skip for all blocks in one contraction (measured slower)
//    5  = 1: folded-skip weight gradient on the generic job table
//    6  = 1: queued generation as per-block launches; = 2: fused chain without the skip waves;
//       = 3: softmax / arg max / emit as three launches; = 4: one launch per head layer
//    7  = 1: print the generation workspace map
//    8  = 1: per-block weight gradients on the generic job table (no wn_wgrad_layer_kernel)
//    9  = 1: no side stream in the weight-gradient phase
//   10  > 0: time splits per utterance of the weight-gradient slabs
//   11  = 1: 128-channel blocks on the exact-fp32 one-kernel forward; = 2: on the two split-precision contractions
//       (gated conv + gate, 1x1 + residual) instead of the streamed-weights one-kernel forward (wn_layer16s.hip)
//   12  = 1: no 256-column wide streamed kernel (N = 256 contractions as two 128-column blocks)
//   13  = 1: 128-channel per-block weight gradients on the generic job table (no wn_wgrad_pair_kernel)
//   14  = 1: global conditioning per block (no single contraction over all blocks)
//   15  = 1: last block's backward without the (zero) output gradient: one-segment product on the fp32 kernel
//   16  = 1: 128-channel blocks: one staged weight-gradient job per tap of the gated conv (du read twice) instead of the
//       transposed-LDS-read kernel with both taps (wn_wgrad_tr.hip); = 2: the staged kernel with both taps (spills, slower)
//   17  = 1: stacks deeper than 1 (layers_per_block > 1): per-call weight gradients instead of the batched job table
//   18  = 1: stacks deeper than 1: training passes on the exact-fp32 composed kernels and exact-fp32 batched weight gradients
//       (no split-precision inner convs with max-abs slots for the inner gradients)
//   19  = 1: head layers' weight gradients on the generic job table (no staged pair jobs)
//   20  = 1: input conv's weight gradients on the generic job table (no dedicated reduction kernel)
//   21  = 1: training passes keep the skip sum and the head's first conv as two steps (no folded V = W_s W_f0 contraction)
//   22  = 1: backward-data chain as two launches per block (no fused g_x(b+1) + g_u(b) kernel)
//   23  = 1: queued generation on the first chain kernel (weights through an LDS image filled by LDS-DMA)
//   25  = 1: generation chain kernel without its L2 helper workgroups
//   26  = 1: the generation pre kernel as its own launch (not inside the previous step's head launch)
//   28  = 1: categorical loss on the one-row-per-wave kernel (no persistent waves with the next row prefetched)
//   27  = 1: the categorical sampling tail of a generation step as its own launch (not inside the head launch);
//       = 2: inside the head launch also for more than 8 utterances
//   24  = 1: the generation chain kernel stamps its phases with s_memtime for blocks 8..11 (wn_debug_gen_ts reads them)
//   30  = 1: the streamed planar contraction (wn_gemm16s.hip) with one row tile per wave instead of two
//   31  = 1: the folded skip contraction on wn_gemm_rows16_kernel (no wn_gemm_planes16s_kernel)
//   32  = 1: the folded skip path's small weight-space products on the 64 x 64-tile kernel (no wn_sgemm_small32_kernel)
//   33  = 1: the conditioning path (mapping Dense stack, conditioning convs) as rows-GEMM launches (no small-product kernel)
//   34  = 1: queued generation of 128-channel blocks on the streamed forward kernel, one launch per block;
//       = 2: on wn_gen_block128_kernel, one launch per block (default: wn_gen_chain128_kernel, all blocks in one launch)
//   36  = 1: the convs of stacks deeper than 1 on the padded rows GEMM (no shifted-plane form of wn_gemm_planes16s_kernel)
//   37  = 1: dW of the head's first conv under the fold (a 1921-long contraction) on wn_wgrad_kernel (no split-K small products)
// thread-local: a caller that switches kernel variants (the range guard's exact-fp32 retry, tools/ A/B runs, tests)
// affects the launches of its own thread only
static thread_local int g_wn_debug[64] = {0};
int wn_debug_get(int key) { return (key >= 0 && key < 64) ? g_wn_debug[key] : 0; }
extern "C" int wn_debug_value(int key) { return wn_debug_get(key); }
extern "C" int wn_debug_set(int key, int value) {
  if (key < 0 || key >= 64) return -1;
  g_wn_debug[key] = value;
  return 0;
}
