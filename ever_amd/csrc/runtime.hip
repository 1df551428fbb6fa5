// Error reporting and ABI identification for libever_hip.so.
#include "common.hpp"
#include <string.h>

namespace evk {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace evk

extern "C" const char* evk_last_error(void) { return evk::g_err; }
extern "C" int evk_abi_version(void) { return 18; }
extern "C" const char* evk_build_arch(void) { return "gfx950"; }
