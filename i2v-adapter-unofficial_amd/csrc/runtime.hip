// Error plumbing and version entry points of libi2v_hip.so.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

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

// The fused sub-block kernels need 144 - 160 KB of dynamic LDS (an opt-in per kernel AND per device) and launch one workgroup per
// CU.  Both facts are cached per (kernel, device) -- not in a function-local static, which would be resolved once per process on
// whichever device happened to be current first: a second device of the same process gets its own opt-in, and a
// refusal is reported by the i2v_*_supported() probes so that callers take the un-fused kernels instead of failing.
int i2v_big_lds_kernel_cus(const void* func, size_t lds_bytes) {
  struct entry { const void* func; int dev; int cus; };
  static std::mutex mu;
  static std::vector<entry> table;
  int dev = 0;
  // (no device at all -- the build container: the probes then answer for the shape alone; a launch would fail on its own)
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  std::lock_guard<std::mutex> lock(mu);
  for (const entry& e : table)
    if (e.func == func && e.dev == dev) return e.cus;
  int cus = 0;
  if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    cus = 0;
  } else if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    cus = 256;
  }
  table.push_back({func, dev, cus});
  return cus;
}

int i2v_persistent_grid(int ntiles, int cus) {
  static const int forced = getenv("I2V_FUSED_TILES_PER_WG") ? atoi(getenv("I2V_FUSED_TILES_PER_WG")) : 0;
  const int per = forced > 0 ? forced : (ntiles + cus - 1) / cus;
  return (ntiles + per - 1) / per;
}

extern "C" const char* i2v_last_error(void) { return g_err; }
extern "C" int i2v_abi_version(void) { return I2V_ABI_VERSION; }
