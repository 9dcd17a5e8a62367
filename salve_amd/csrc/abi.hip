// abi.hip -- version and error reporting of the C ABI (include/salve_hip.h).
#include <stdio.h>
#include <string.h>

#include "../../include/salve_hip.h"
#include "salve_common.h"

static thread_local char g_err[512] = "";

bool salve_fail(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return false;
}

bool salve_fail_hip(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return false;
}

extern "C" {
int salve_hip_version(void) { return SALVE_HIP_ABI_VERSION; }
const char* salve_last_error(void) { return g_err; }
}
