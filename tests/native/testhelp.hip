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
// box").  Grid-stride over 16-byte lanes, U independent loads in flight per thread before the stores; NT: non-temporal loads and stores
// (a copy re-uses nothing: the lines need not stay in the L2 / memory-side cache).
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy16_kernel(f32x4* __restrict__ dst, const f32x4* __restrict__ src, long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

// mode: bit 0 = non-temporal accesses, bit 1 = 8 instead of 4 loads in flight per thread
extern "C" int salve_debug_copy16(void* dst, const void* src, long long n16, int32_t blocks, int32_t mode, void* stream) {
    if (!dst || !src || n16 <= 0 || blocks <= 0) return -1;
    f32x4* d = reinterpret_cast<f32x4*>(dst);
    const f32x4* s = reinterpret_cast<const f32x4*>(src);
    hipStream_t st = (hipStream_t)stream;
    switch (mode & 3) {
        case 0: hipLaunchKernelGGL((copy16_kernel<4, false>), dim3(blocks), dim3(256), 0, st, d, s, n16); break;
        case 1: hipLaunchKernelGGL((copy16_kernel<4, true>), dim3(blocks), dim3(256), 0, st, d, s, n16); break;
        case 2: hipLaunchKernelGGL((copy16_kernel<8, false>), dim3(blocks), dim3(256), 0, st, d, s, n16); break;
        default: hipLaunchKernelGGL((copy16_kernel<8, true>), dim3(blocks), dim3(256), 0, st, d, s, n16); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int salve_debug_burn(int32_t blocks, int32_t iters, int32_t mode, float* sink, void* stream) {
    hipLaunchKernelGGL(burn_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, mode, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
