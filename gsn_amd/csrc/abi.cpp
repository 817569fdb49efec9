// Error reporting, version and device discovery for libgsn_hip.so (C ABI: gsn_last_error, gsn_version, gsn_device_count).
#include <hip/hip_runtime.h>

#include <cstring>

#include "gsn_internal.h"

namespace gsn {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace gsn

extern "C" const char *gsn_last_error(void) { return gsn::g_err; }

extern "C" int gsn_version(void) { return GSN_ABI_VERSION; }

extern "C" int gsn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int good = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
    }
    return good;
}

namespace gsn {
int current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}
}  // namespace gsn
