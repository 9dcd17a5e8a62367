// stream_pieces.hip -- stand-alone: what a mixed read + write stream reaches on MI355X when it walks [pixels][C] fp16 tensors in
// COLUMN PIECES of P bytes per row (the access shape of expand_chain_kernel: 128-row tiles, one 32-channel = 64-byte chunk after
// the other) against wider pieces and against whole rows.  256 persistent workgroups of 512 threads, 8 x 16-byte loads in flight
// per lane, every loaded piece is stored to the same place of a second tensor.
//   hipcc --offload-arch=gfx950 -O3 -o stream_pieces tools/probe/repro/stream_pieces.hip && ./stream_pieces
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int P>   // bytes per row piece
__global__ __launch_bounds__(512) void copy_pieces(const uint4* __restrict__ x, uint4* __restrict__ y, int n_tiles, int row_bytes) {
    constexpr int LPR = P / 16;            // lanes per row piece
    constexpr int RPP = 512 / LPR;         // rows per pass of the workgroup
    const int lane_col = threadIdx.x % LPR, lane_row = threadIdx.x / LPR;
    const int pieces = row_bytes / P;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t base = (size_t)t * 128 * row_bytes / 16;    // uint4 units
        for (int pc = 0; pc < pieces; pc++) {
            // one piece of the 128-row tile: 128 * P bytes; the workgroup covers RPP rows per pass
            uint4 v[8];
            int n = 0;
#pragma unroll
            for (int r0 = 0; r0 < 128; r0 += RPP) {
                if (n < 8) v[n++] = x[base + (size_t)(r0 + lane_row) * (row_bytes / 16) + pc * LPR + lane_col];
            }
            n = 0;
#pragma unroll
            for (int r0 = 0; r0 < 128; r0 += RPP) {
                if (n < 8) y[base + (size_t)(r0 + lane_row) * (row_bytes / 16) + pc * LPR + lane_col] = v[n++];
            }
        }
    }
}

template <int P>
float run(const uint4* x, uint4* y, int n_tiles, int row_bytes) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(copy_pieces<P>, dim3(256), dim3(512), 0, 0, x, y, n_tiles, row_bytes);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(copy_pieces<P>, dim3(256), dim3(512), 0, 0, x, y, n_tiles, row_bytes);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    for (int row_bytes : {1024, 2048}) {     // C4 = 512 / 1024 channels
        const size_t M = (size_t)4096 * (row_bytes == 1024 ? 784 : 196);
        const size_t bytes = M * row_bytes;
        uint4 *x, *y;
        hipMalloc(&x, bytes); hipMalloc(&y, bytes);
        hipMemset(x, 1, bytes);
        const int n_tiles = (int)(M / 128);
        const float t64 = run<64>(x, y, n_tiles, row_bytes), t128 = run<128>(x, y, n_tiles, row_bytes), t256 = run<256>(x, y, n_tiles, row_bytes),
                    t1024 = run<1024>(x, y, n_tiles, row_bytes);
        printf("rows of %d bytes, %.2f GB read + %.2f GB written: 64-byte pieces %.0f GB/s, 128-byte %.0f, 256-byte %.0f, 1024-byte %.0f\n", row_bytes,
               bytes / 1e9, bytes / 1e9, 2 * bytes / t64 / 1e6, 2 * bytes / t128 / 1e6, 2 * bytes / t256 / 1e6, 2 * bytes / t1024 / 1e6);
        hipFree(x); hipFree(y);
    }
    return 0;
}
