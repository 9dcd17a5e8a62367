// testhelp.hip -- TEST helper library (tests/native/libsalve_testhelp.so, built by __graft_entry__.build()): synthetic load
// kernels used to study how the rasteriser behaves next to other work on the same compute units.  Not part of the product:
// libsalve_hip.so neither contains nor exports any of this.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "testhelp.h"

namespace {
typedef __attribute__((__ext_vector_type__(8))) __bf16 bf16x8;
typedef __attribute__((__ext_vector_type__(4))) float f32x4;

// mode 0: MFMA only (registers), 1: VALU fp32 only, 2: LDS reads only, 3: MFMA + LDS reads
__global__ __launch_bounds__(256) void burn_kernel(int iters, int mode, float* sink) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[16384];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = (uint16_t)(i * 7 + 3);
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (__bf16)(float)(tid + k); b[k] = (__bf16)(float)(k + 1); }
    float v = (float)tid;
    for (int it = 0; it < iters; it++) {
        if (mode == 0 || mode == 3) {
#pragma unroll
            for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        if (mode == 1) {
#pragma unroll
            for (int u = 0; u < 64; u++) v = v * 1.0001f + 0.5f;
        }
        if (mode == 2 || mode == 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(lds + ((tid * 8 + u * 2048 + it * 8) & 16376));
                a[u & 7] = t[u & 7];
            }
        }
    }
    if (acc[0] + v == 12345.678f) sink[0] = acc[0];
}

// float4 device copy: the HBM copy microbenchmark of bench.py ("confirm the HBM peak with a copy microbenchmark on the
// box").  Grid-stride over 16-byte lanes, four independent loads in flight per thread before the stores.
__global__ __launch_bounds__(256) void copy16_kernel(float4* __restrict__ dst, const float4* __restrict__ src, long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int salve_debug_copy16(void* dst, const void* src, long long n16, int32_t blocks, void* stream) {
    if (!dst || !src || n16 <= 0 || blocks <= 0) return -1;
    hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(dst),
                       reinterpret_cast<const float4*>(src), n16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int salve_debug_burn(int32_t blocks, int32_t iters, int32_t mode, float* sink, void* stream) {
    hipLaunchKernelGGL(burn_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, mode, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
