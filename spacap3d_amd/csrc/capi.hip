// Library-level entry points of libspacap_hip.so: version, error text, device probe, and the
// reference's launch-size helper restated for the host side.
#include <math.h>
#include <stdarg.h>

#include "common.hpp"

namespace spacap {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace spacap

extern "C" int spacap_abi_version(void) { return SPACAP_ABI_VERSION; }

extern "C" const char *spacap_last_error(void) { return spacap::g_err; }

extern "C" int spacap_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    spacap::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return SPACAP_E_NO_DEVICE;
  }
  return n;
}

// include/cuda_utils.h:15-19 of the reference: pow_2 = (int)(log(w) / log(2)); clamp(1 << pow_2, 1, 512).
extern "C" int spacap_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}
