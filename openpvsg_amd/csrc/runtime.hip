// Error reporting + library identification for the C-ABI (include/openpvsg_hip.h).
#include "common.h"
#include <stdarg.h>

namespace pvsg {
static thread_local char g_err[512] = {0};
char* err_buf() { return g_err; }
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace pvsg

extern "C" const char* pvsg_last_error(void) { return pvsg::err_buf(); }
extern "C" const char* pvsg_version(void) { return "openpvsg_amd-hip 0.1 (gfx950)"; }
extern "C" int pvsg_abi_version(void) { return 6; }
