// Thread-local error string + version for the C ABI (include/laplace_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "laplace_hip.h"

namespace lk {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lk

extern "C" int lk_version(void) { return 100; }
extern "C" const char* lk_last_error(void) { return lk::g_err; }
