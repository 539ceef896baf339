// abi.hip — version / error plumbing of the C ABI (include/gnnops.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void gnnops_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int gnnops_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gnnops_set_error("%s: %s", what, hipGetErrorString(e));
        return GNNOPS_ELAUNCH;
    }
    return GNNOPS_OK;
}

extern "C" int gnnops_version(void) { return GNNOPS_ABI_VERSION; }
extern "C" const char* gnnops_last_error(void) { return g_err; }
