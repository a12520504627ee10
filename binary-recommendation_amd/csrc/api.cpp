// Error string, version and device query of the C-ABI (host only).
#include "common.h"
#include <string.h>

namespace br {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
static thread_local const StepStateDev* g_step_state = nullptr;
const StepStateDev* current_step_state() { return g_step_state; }
void set_current_step_state(const StepStateDev* p) { g_step_state = p; }
}  // namespace br

extern "C" const char* brGetLastError(void) { return br::g_err; }
extern "C" int brVersion(void) { return 100; }

extern "C" int brDeviceInfo(int* cu_count, int* wave_size, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    br::set_error("brDeviceInfo: hipGetDevice: %s", hipGetErrorString(e));
    return BR_ERR_HIP;
  }
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    br::set_error("brDeviceInfo: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return BR_ERR_HIP;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return BR_OK;
}
