// C-ABI runtime helpers: error string, version, device probe.  See include/ullsam_hip.h.
#include "common.h"
#include "ullsam_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void ullsam_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ullsam_last_error_string(void) { return g_err; }
extern "C" int ullsam_abi_version(void) { return ULLSAM_ABI_VERSION; }

// Returns the number of visible HIP devices (0 when none); never throws.
extern "C" int ullsam_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
