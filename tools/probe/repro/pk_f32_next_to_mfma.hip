// Reproducer attempt for DESIGN.md section 8: do packed fp32 VALU ops (v_pk_mul_f32 / v_pk_add_f32) return wrong results
// when waves of ANOTHER kernel issue MFMAs on the same CU?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk pk_f32_next_to_mfma.hip
// Prints mismatches of the packed chain against the scalar chain, alone and with the MFMA kernel running on a second stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__ext_vector_type__(2))) float f2;
typedef __attribute__((__ext_vector_type__(8))) _Float16 h8;
typedef __attribute__((__ext_vector_type__(4))) float f4;
__global__ __launch_bounds__(256) void mfma_burn(int iters, float* sink) {
    f4 acc = {0, 0, 0, 0};
    h8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(float)(threadIdx.x + k); b[k] = (_Float16)(float)(k + 1); }
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}
__global__ __launch_bounds__(256) void pk_chain(int iters, unsigned* bad) {
    const float s = (float)(blockIdx.x * 256 + threadIdx.x);
    f2 x = {1.0f + s * 1e-6f, 2.0f - s * 1e-6f};
    float y0 = x[0], y1 = x[1];
    const f2 c1 = {1.0001f, 0.9999f}, c2 = {0.5f, -0.25f};
    for (int it = 0; it < iters; it++) {
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c1));    // packed chain
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(c2));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(y0), "v"(c1[0]));  // scalar chain, same operations
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(y1), "v"(c1[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(y0) : "v"(y0), "v"(c2[0]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(y1) : "v"(y1), "v"(c2[1]));
        if ((it & 63) == 63) { x = x * 0.5f; y0 *= 0.5f; y1 *= 0.5f; }
    }
    if (__float_as_uint(x[0]) != __float_as_uint(y0) || __float_as_uint(x[1]) != __float_as_uint(y1)) atomicAdd(bad, 1u);
}
int main() {
    unsigned* bad; float* sink;
    hipMalloc(&bad, 4); hipMalloc(&sink, 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int with_mfma = 0; with_mfma < 2; with_mfma++) {
        unsigned total = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipMemsetAsync(bad, 0, 4, s1);
            hipStreamSynchronize(s1);
            if (with_mfma) hipLaunchKernelGGL(mfma_burn, dim3(2048), dim3(256), 0, s2, 20000, sink);
            hipLaunchKernelGGL(pk_chain, dim3(4096), dim3(256), 0, s1, 20000, bad);
            hipDeviceSynchronize();
            unsigned h = 0;
            hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            total += h;
        }
        printf("packed fp32 chain %s: %u mismatching threads in 20 x 1,048,576\n", with_mfma ? "next to the MFMA kernel" : "alone", total);
    }
    return 0;
}
