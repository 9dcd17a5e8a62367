// Follow-up to pk_f32_next_to_mfma.hip (packed fp32 alone is sound).  The SLP-vectorised build of bev_render.hip feeds the
// result of a TRANSCENDENTAL instruction into a PACKED one three instructions later:
//     v_sqrt_f32 v33, v77 ; s_mov_b32 ; v_cmp_gt_f32 ; v_mul_f32 v10, ... ; v_pk_mul_f32 v[20:21], v[32:33], v[14:15]
// and that build miscomputes (round 1: wrong pixels next to MFMA kernels; round 2: a fault).  This program issues the pair by
// hand (inline asm: hipcc inserts no hazard wait states into an asm statement) with 0..4 independent VALU instructions in
// between, and counts results that differ from  s * c.x , sqrt(s) * c.y  computed with plain instructions -- alone and with
// an MFMA-only kernel on a second stream.     hipcc --offload-arch=gfx950 -O3 -o tp trans_then_pk_f32.hip
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
#define FILL1 "v_add_f32 v44, v44, v44\n"
#define FILL0
#define FILL2 FILL1 FILL1
#define FILL3 FILL2 FILL1
#define FILL4 FILL3 FILL1
#define CHAIN(NAME, FILL)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(int iters, unsigned* bad) {                                            \
        float s = 1.0f + (float)(blockIdx.x * 256 + threadIdx.x) * 1e-3f;                                             \
        const f2 c = {1.25f, 0.75f};                                                                                   \
        unsigned wrong = 0;                                                                                            \
        for (int it = 0; it < iters; it++) {                                                                           \
            float lo, hi, r;                                                                                           \
            asm volatile("v_mov_b32 v40, %2\n"                                                                         \
                         "s_nop 4\n"                                                                                   \
                         "v_sqrt_f32 v41, %2\n" FILL "v_pk_mul_f32 v[42:43], v[40:41], %3\n"                           \
                         "s_nop 4\n"                                                                                   \
                         "v_mov_b32 %0, v42\n"                                                                         \
                         "v_mov_b32 %1, v43\n"                                                                         \
                         : "=v"(lo), "=v"(hi) : "v"(s), "v"(c) : "v40", "v41", "v42", "v43", "v44");                   \
            asm volatile("v_sqrt_f32 %0, %1\n s_nop 4" : "=v"(r) : "v"(s));                                           \
            const float e0 = s * c[0], e1 = r * c[1];                                                                  \
            if (__float_as_uint(lo) != __float_as_uint(e0) || __float_as_uint(hi) != __float_as_uint(e1)) wrong++;     \
            s = s * 1.0001f + 0.001f;                                                                                  \
            if (s > 1e6f) s = 1.0f;                                                                                    \
        }                                                                                                              \
        if (wrong) atomicAdd(bad, wrong);                                                                              \
    }
CHAIN(chain0, FILL0)
CHAIN(chain1, FILL1)
CHAIN(chain2, FILL2)
CHAIN(chain3, FILL3)
CHAIN(chain4, FILL4)
int main() {
    unsigned* bad; float* sink;
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&sink, 4);
    hipStream_t s1, s2;
    (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    void (*kern[5])(int, unsigned*) = {chain0, chain1, chain2, chain3, chain4};
    for (int gap = 0; gap < 5; gap++)
        for (int with_mfma = 0; with_mfma < 2; with_mfma++) {
            unsigned long long total = 0;
            for (int rep = 0; rep < 5; rep++) {
                (void)hipMemsetAsync(bad, 0, 4, s1);
                (void)hipStreamSynchronize(s1);
                if (with_mfma) hipLaunchKernelGGL(mfma_burn, dim3(2048), dim3(256), 0, s2, 4000, sink);
                hipLaunchKernelGGL(kern[gap], dim3(4096), dim3(256), 0, s1, 2000, bad);
                (void)hipDeviceSynchronize();
                unsigned h = 0;
                (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
                total += h;
            }
            printf("v_sqrt_f32 -> %d VALU -> v_pk_mul_f32, %s: %llu wrong results of %llu\n", gap, with_mfma ? "next to MFMA kernel" : "alone           ",
                   total, 5ull * 4096 * 256 * 2000);
        }
    return 0;
}
