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
static int g_wn_debug[16] = {0};
int wn_debug_get(int key) { return (key >= 0 && key < 16) ? g_wn_debug[key] : 0; }
extern "C" int wn_debug_set(int key, int value) {
  if (key < 0 || key >= 16) return -1;
  g_wn_debug[key] = value;
  return 0;
}
