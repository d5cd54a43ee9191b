// Error plumbing + version for libacr_hip.so (see include/acr_hip.h for the ABI contract).
#include <stdarg.h>

#include "acr_common.h"

static thread_local char g_err[512] = "";

void acr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int acr_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        acr_set_error("%s: %s", what, hipGetErrorString(e));
        return ACR_ERR_LAUNCH;
    }
    return ACR_OK;
}

extern "C" int acr_version(void) { return ACR_ABI_VERSION; }
extern "C" const char* acr_last_error(void) { return g_err; }
