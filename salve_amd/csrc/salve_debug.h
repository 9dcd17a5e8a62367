// salve_debug.h -- development entry points of libsalve_hip.so that are NOT part of the product ABI (include/salve_hip.h):
// tools/ bind them by name.
#pragma once
#include <stdint.h>
extern "C" {
// A synthetic load kernel -- mode 0 MFMA only, 1 VALU only, 2 LDS reads only, 3 MFMA + LDS -- used to study co-residency
// with the rasteriser (tools/debug_overlap2.py, tests/test_gpu_facade.py: the packed-fp32 regression test).
int salve_debug_burn(int32_t blocks, int32_t iters, int32_t mode, float* sink, void* stream);
}
