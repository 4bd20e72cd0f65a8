// ABI bookkeeping for libunopose_hip.so: version + last-error text.
#include <stdarg.h>

#include "common.h"

namespace unopose {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace unopose

extern "C" {
int unopose_abi_version(void) { return 1; }
const char *unopose_last_error(void) { return unopose::g_err; }
}
