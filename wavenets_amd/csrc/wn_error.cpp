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

// Per-thread switches of the launchers (wn_debug_set / wn_debug_value in the C-ABI).  Round 4 retired the A/B knobs of the
// kernel variants that lost (DESIGN.md section 10 lists what was removed with them); what is left:
//    1  = 1: exact-fp32 MFMA kernels everywhere (no fp16 hi|lo operand split): the mode a range-guard trip repeats a pass
//            in, and the second arithmetic every parity test runs
//    2  = 1: queued generation of 128-channel blocks in ONE workgroup per utterance tile (wn_gen_chain128_kernel) instead
//            of the relay over one workgroup per block (wn_gen_relay128_kernel): the cross-check of the relay's hand-offs
//    3  = b + 1: fault injection for the relay's watchdog: block b withholds its hand-over, its successor gives up after
//            its bounded wait and wn_generate's watchdog word reports it (tests only)
//    9  = 1: no side stream in the weight-gradient phase (everything on the caller's stream)
//   24  = 1: the generation chain kernel stamps its phases with s_memtime for blocks 8..11 (wn_debug_gen_ts reads them)
//   29  > 0: timing ablations of the 128-channel block forward; only in -DWN_S128_DIAG builds (tools/time_s128.py)
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
