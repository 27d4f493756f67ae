// common.hip — error reporting, version, device query.
#include "mpn_internal.h"

namespace mpn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mpn

extern "C" int mpn_version(void) { return MPN_VERSION; }
extern "C" const char *mpn_last_error(void) { return mpn::g_err; }

extern "C" int mpn_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes) {
  int dev = 0;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MPN_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return MPN_OK;
}
