// Error convention + version of the C-ABI (include/bm_hip.h).
#include "bm_common.h"
#include <stdarg.h>

thread_local char bm_err_buf[512] = {0};

int bm_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(bm_err_buf, sizeof(bm_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

int bm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bm_set_error((int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
    return BM_OK;
}

extern "C" const char* bm_last_error(void) { return bm_err_buf; }
extern "C" int bm_version(void) { return 100; }   // 0.1.0

// Number of HIP devices visible; <0 on error.  Lets the Python side fail loudly early.
extern "C" int bm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { bm_set_error((int)e, "hipGetDeviceCount: %s", hipGetErrorString(e)); return -1; }
    return n;
}
