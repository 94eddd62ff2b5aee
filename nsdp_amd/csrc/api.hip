// ABI bookkeeping for libnsdp_hip.so: version, thread-local last-error string, device probe.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace nsdp {
static thread_local char g_last_error[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}
}  // namespace nsdp

extern "C" {

int nsdp_abi_version(void) { return 1; }

const char *nsdp_last_error(void) { return nsdp::g_last_error; }

int nsdp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
