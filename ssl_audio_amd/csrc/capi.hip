// Error storage + device info for the C ABI (include/ssl_audio_hip.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ssl_audio_hip.h"

static thread_local char g_err[512] = "";

extern "C" void sa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sa_last_error(void) { return g_err; }

extern "C" int sa_abi_version(void) { return 6; }

extern "C" int sa_device_info(char* name_host, int name_len, int* cu_count_host) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    sa_set_error("sa_device_info: no HIP device");
    return 2;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    sa_set_error("sa_device_info: hipGetDeviceProperties failed");
    return 2;
  }
  if (name_host && name_len > 0) {
    strncpy(name_host, prop.gcnArchName, (size_t)name_len - 1);
    name_host[name_len - 1] = 0;
  }
  if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
  return 0;
}
