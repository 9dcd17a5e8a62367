// salve_common.h -- error plumbing shared by the translation units of libsalve_hip.so.
#pragma once
#include <hip/hip_runtime.h>

// Records a thread-local message for salve_last_error() and returns false.
bool salve_fail(const char* msg);
bool salve_fail_hip(const char* what, hipError_t e);

#define SALVE_HIP_CHECK(expr)                                   \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) {                                 \
            salve_fail_hip(#expr, _e);                          \
            return SALVE_ERR_HIP;                               \
        }                                                       \
    } while (0)
