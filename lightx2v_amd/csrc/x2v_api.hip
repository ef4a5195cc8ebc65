// Host-side plumbing of the C-ABI: thread-local error string, device check, version.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "x2v_common.h"

namespace x2v {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return X2V_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return X2V_E_HIP;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: remember what was raised per (kernel, device), not per process.
int ensure_dynamic_lds(const void* kernel, int bytes, const char* what) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> raised;
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), what);
  if (rc != X2V_OK) return rc;
  std::lock_guard<std::mutex> lock(mu);
  int& have = raised[std::make_pair(kernel, dev)];
  if (have >= bytes) return X2V_OK;
  rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
  if (rc == X2V_OK) have = bytes;
  return rc;
}

}  // namespace x2v

using namespace x2v;

extern "C" __attribute__((visibility("default"))) const char* x2v_last_error(void) { return g_err; }

extern "C" __attribute__((visibility("default"))) const char* x2v_version(void) { return "x2v-hip 0.1.0 (gfx950)"; }

extern "C" __attribute__((visibility("default"))) int x2v_switches(char* buf, int buf_len) {
  X2V_REQUIRE(buf != nullptr && buf_len > 0, X2V_E_ARG, "x2v_switches: no buffer");
  const int n = snprintf(buf, (size_t)buf_len, "X2V_GEMM_CONTINUOUS=%d X2V_GEMM_FP8_CONTINUOUS=%d X2V_ATTN_MAP=%d X2V_ATTN_ROT=%d", gemm_continuous_switch(),
                         gemm_fp8_continuous_switch(), attn_map_switch(), attn_rot_switch());
  X2V_REQUIRE(n > 0 && n < buf_len, X2V_E_ARG, "x2v_switches: buffer of %d bytes too small", buf_len);
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len) {
  hipDeviceProp_t prop;
  int rc = check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
  if (rc != X2V_OK) return rc;
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, prop.gcnArchName, (size_t)arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_init(int device) {
  int n = 0;
  int rc = check_hip(hipGetDeviceCount(&n), "hipGetDeviceCount");
  if (rc != X2V_OK) return rc;
  X2V_REQUIRE(device >= 0 && device < n, X2V_E_ARG, "x2v_init: device %d out of range (%d visible)", device, n);
  hipDeviceProp_t prop;
  rc = check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
  if (rc != X2V_OK) return rc;
  X2V_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, X2V_E_ARCH, "x2v_init: device %d is %s; this library is built for gfx950 only", device,
              prop.gcnArchName);
  return X2V_OK;
}
