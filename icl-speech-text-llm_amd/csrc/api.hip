// api.hip — library-level entry points and error plumbing of libicl_hip.
#include "common.h"

static thread_local char g_err[512] = "";

void icl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int icl_abi_version(void) { return ICL_ABI_VERSION; }
extern "C" const char* icl_last_error(void) { return g_err; }
extern "C" int icl_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}
