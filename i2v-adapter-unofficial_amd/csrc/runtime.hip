// Error plumbing and version entry points of libi2v_hip.so.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void i2v_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int i2v_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    i2v_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return I2V_ERR_LAUNCH;
  }
  return I2V_OK;
}

extern "C" const char* i2v_last_error(void) { return g_err; }
extern "C" int i2v_abi_version(void) { return I2V_ABI_VERSION; }
