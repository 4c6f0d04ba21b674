// C-ABI housekeeping: error string, version, device probe.
#include "mi_common.h"
#include <string.h>

namespace mi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
  return MI_ERR_HIP;
}

}  // namespace mi

extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }

extern "C" const char* mi_last_error(void) { return mi::g_err; }

extern "C" int mi_device_supported(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return mi::hip_fail(e, "hipGetDevice");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return mi::hip_fail(e, "hipGetDeviceProperties");
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
