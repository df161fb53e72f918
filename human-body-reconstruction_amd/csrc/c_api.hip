// Library-level entry points of libhbr_hip.so (version, error strings, device probe).
#include <string.h>

#include "hbr_common.h"

extern "C" int hbr_version(void) { return HBR_VERSION; }

extern "C" const char* hbr_strerror(int code) {
  switch (code) {
    case HBR_OK: return "ok";
    case HBR_EINVAL: return "invalid argument (null pointer, negative size, misaligned or inconsistent shape)";
    case HBR_EUNSUPPORTED: return "configuration not supported by the gfx950 kernels";
    case HBR_ELAUNCH: return "HIP kernel launch failed";
    case HBR_EWORKSPACE: return "workspace too small";
  }
  return "unknown error";
}

extern "C" int hbr_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
