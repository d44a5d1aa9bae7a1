// Error plumbing, version and device checks of libvbx_hip.so.
#include "common.hpp"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vbx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vbx_last_error(void) { return g_err; }
extern "C" int vbx_version(void) { return VBX_VERSION; }

extern "C" int vbx_check_device(int dev) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    vbx_set_error("hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
    return (int)e;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    vbx_set_error("device %d is %s; libvbx_hip is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return VBX_EUNSUPPORTED;
  }
  return 0;
}
