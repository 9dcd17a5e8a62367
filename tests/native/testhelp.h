// testhelp.h -- entry points of the TEST helper library tests/native/libsalve_testhelp.so (not part of the product ABI,
// include/salve_hip.h, and not in libsalve_hip.so); tests and tools/ bind them by name with ctypes.
#pragma once
#include <stdint.h>
extern "C" {
// A synthetic load kernel -- mode 0 MFMA only, 1 VALU only, 2 LDS reads only, 3 MFMA + LDS -- used to study co-residency
// with the rasteriser (tools/debug_overlap2.py, tests/test_gpu_rasteriser.py: the packed-fp32 regression test).
int salve_debug_burn(int32_t blocks, int32_t iters, int32_t mode, float* sink, void* stream);
// float4 device copy of n16 16-byte lanes on `blocks` workgroups of 256 threads: bench.py's HBM copy microbenchmark (measurement only).
// mode: bit 0 = non-temporal loads / stores, bit 1 = 8 (instead of 4) loads in flight per thread.
int salve_debug_copy16(void* dst, const void* src, long long n16, int32_t blocks, int32_t mode, void* stream);
}
