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
