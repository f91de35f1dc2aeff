// api.hip -- library info / error string of libmeld_hip.so
#include "common.hpp"

#include <stdarg.h>

namespace meld {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace meld

extern "C" int meld_abi_version(void) { return 1; }
extern "C" const char* meld_last_error(void) { return meld::err_buf(); }
extern "C" int meld_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return MELD_ERR_HIP;
  return n;
}
