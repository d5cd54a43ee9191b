// Error plumbing + version for libacr_hip.so (see include/acr_hip.h for the ABI contract).
#include <stdarg.h>

#include <atomic>

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

// ---- explicit option table (the library's only process-wide state; see include/acr_hip.h) ------------------
static std::atomic<int32_t> g_opt[ACR_OPT_COUNT_] = {{2}, {0}, {0}, {2}, {8}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {3}};

int32_t acr_opt(int option) { return g_opt[option].load(std::memory_order_relaxed); }

extern "C" int acr_set_option(int32_t option, int32_t value) {
    ACR_CHECK_ARG(option >= 0 && option < ACR_OPT_COUNT_, "acr_set_option: unknown option %d", option);
    g_opt[option].store(value, std::memory_order_relaxed);
    return ACR_OK;
}

extern "C" int32_t acr_get_option(int32_t option) {
    if (option < 0 || option >= ACR_OPT_COUNT_) return INT32_MIN;
    return acr_opt(option);
}
