// Error plumbing + version for the C-ABI (thread-local error string; no C++ exceptions cross the ABI).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void tatt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* tatt_last_error(void) { return g_err; }
int tatt_version(void) { return 100; }
}
