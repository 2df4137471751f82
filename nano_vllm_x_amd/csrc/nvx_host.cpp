// Host-side plumbing shared by all entry points: error text, ABI version.
#include "nvx_dev.h"

thread_local char nvx_err_buf[512] = "";

int nvx_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(nvx_err_buf, sizeof(nvx_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

int nvx_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvx_fail(NVX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NVX_OK;
}

extern "C" int nvx_abi_version(void) { return NVX_ABI_VERSION; }
extern "C" const char* nvx_last_error(void) { return nvx_err_buf; }
