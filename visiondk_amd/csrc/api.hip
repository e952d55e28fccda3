// api.hip — error reporting and library identification for the C ABI (include/visiondk.h).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include "vdk_host.h"

static thread_local char g_err[512] = "";

int vdk_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "error");
  return code;
}

int vdk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return VDK_ELAUNCH;
  }
  return VDK_OK;
}

extern "C" {

const char* vdk_last_error(void) { return g_err; }

// 1 when this shared object was compiled by hipcc for gfx950; the CPU SIMT emulation used by the
// test-suite (tests/emu) reports 0 so that product code can refuse to run on it.
int vdk_is_device_build(void) {
#ifdef VDK_EMU
  return 0;
#else
  return 1;
#endif
}

int vdk_abi_version(void) { return 1; }

}  // extern "C"
