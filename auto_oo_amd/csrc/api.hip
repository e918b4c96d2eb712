// Library-level entry points of liboovqe_hip.so (include/oovqe.h).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void oovqe_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int oovqe_version(void) { return 100; }

extern "C" const char* oovqe_last_error(void) { return g_err; }

extern "C" int oovqe_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        oovqe_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}
