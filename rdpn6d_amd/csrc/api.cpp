// Error channel + misc entry points of the C ABI (no kernels here).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/rdpn6d.h"

static thread_local char g_err[512] = "";

extern "C" void rdpn6d_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rdpn6d_last_error(void) { return g_err; }
extern "C" int rdpn6d_version(void) { return 100; }

extern "C" int rdpn6d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
