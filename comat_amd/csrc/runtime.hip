// runtime.hip — error reporting and ABI version for libcomat_hip.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void comat_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int comat_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        comat_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return COMAT_ELAUNCH;
    }
    return COMAT_OK;
}

extern "C" int comat_abi_version(void) { return COMAT_ABI_VERSION; }
extern "C" const char* comat_last_error(void) { return g_err; }
