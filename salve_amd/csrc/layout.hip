// layout.hip -- rasterised-LAYOUT modality for gfx950: filled room polygon + thick anti-aliased W/D/O segments per image.
//
// Stands behind salve/utils/bev_rendering_utils.py:104-251 (rasterize_single_layout: cv2.fillPoly :159-179, cv2.line with
// LINE_AA :210-251; np.flipud :155).  All pixel arithmetic is integer and follows oracle/layout_oracle.py rule for rule (the
// OpenCV arithmetic itself is "parity unpinned": cv2 is not in the image -- see that file's header):
//   * polygon: even-odd scanline fill on 16.16 fixed-point edge crossings (an edge covers y0 <= y < y1, a span runs from
//     ceil(left) to floor(right)) united with the 8-connected LineIterator pixels of every edge, in closed form;
//   * segment: capsule of radius thickness / 2, coverage clamp(thickness / 2 + 1/2 - distance, 0, 1) in 1/256 steps from the
//     integer square root of the exact squared distance, blended dst += ((colour - dst) * coverage + 128) >> 8, in list order.
// One thread per pixel; the few dozen edges of an image are read by every thread of it (scalar loads, L2 / K$ resident).
// The work is tiny next to the texture rasteriser (one launch renders thousands of layouts); no LDS, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/salve_hip.h"
#include "salve_common.h"

namespace {

__device__ __forceinline__ long long isqrt64(long long v) {
    if (v <= 0) return 0;
    long long r = (long long)sqrt((double)v);
    while (r * r > v) r--;
    while ((r + 1) * (r + 1) <= v) r++;
    return r;
}

// floor division (the oracle's Python `//`) for a positive divisor
__device__ __forceinline__ long long floordiv(long long a, long long b) {
    long long q = a / b;
    if ((a % b != 0) && (a < 0)) q--;
    return q;
}

__device__ __forceinline__ bool on_line8(int x, int y, int x1, int y1, int x2, int y2) {
    const int dx = abs(x2 - x1), dy = abs(y2 - y1);
    const int sx = x2 >= x1 ? 1 : -1, sy = y2 >= y1 ? 1 : -1;
    if (dy > dx) {
        const int j = (y - y1) * sy;
        if (j < 0 || j > dy) return false;
        const long long m = j == 0 ? 0 : max(0ll, -floordiv(-(2ll * dx * j - dy), 2ll * dy));
        return (long long)((x - x1) * sx) == m;
    }
    const int j = (x - x1) * sx;
    if (j < 0 || j > dx) return false;
    if (dx == 0) return y == y1;
    const long long m = j == 0 ? 0 : max(0ll, -floordiv(-(2ll * dy * j - dx), 2ll * dx));
    return (long long)((y - y1) * sy) == m;
}

__global__ __launch_bounds__(256) void layout_kernel(const salve_layout_t* __restrict__ layouts, const int32_t* __restrict__ poly_xy,
                                                     const int32_t* __restrict__ segs, int H, int W, uint32_t* __restrict__ out) {
    const int img = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const salve_layout_t L = layouts[img];
    const int32_t* P = poly_xy + 2 * (size_t)L.poly_off;
    int r = 0, g = 0, b = 0;
    if (L.n_poly > 0) {
        int le = 0, lt = 0;
        bool inside = false;
        const long long xf = (long long)x << 16;
        for (int i = 0; i < L.n_poly; i++) {
            const int k = i + 1 == L.n_poly ? 0 : i + 1;
            int x0 = P[2 * i], y0 = P[2 * i + 1], x1 = P[2 * k], y1 = P[2 * k + 1];
            inside |= on_line8(x, y, x0, y0, x1, y1);
            if (y0 == y1) continue;
            if (y0 > y1) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
            if (y0 <= y && y < y1) {
                const long long num = (long long)(x1 - x0) << 16;
                const long long slope = num / (y1 - y0);                       // C division truncates toward zero
                const long long c = ((long long)x0 << 16) + (long long)(y - y0) * slope;
                le += c <= xf;
                lt += c < xf;
            }
        }
        if (inside || (le & 1) || (lt & 1)) r = g = b = 255;
    }
    const int32_t* S = segs + 8 * (size_t)L.seg_off;
    for (int i = 0; i < L.n_seg; i++) {
        const int x1 = S[8 * i], y1 = S[8 * i + 1], x2 = S[8 * i + 2], y2 = S[8 * i + 3];
        const uint32_t col = (uint32_t)S[8 * i + 4];
        const int thickness = S[8 * i + 5];
        const long long vx = x2 - x1, vy = y2 - y1, wx = x - x1, wy = y - y1;
        const long long L2 = vx * vx + vy * vy, dot = wx * vx + wy * vy;
        long long d256;
        if (L2 == 0 || dot <= 0) {
            d256 = isqrt64((wx * wx + wy * wy) << 16);
        } else if (dot >= L2) {
            const long long ux = x - x2, uy = y - y2;
            d256 = isqrt64((ux * ux + uy * uy) << 16);
        } else {
            const long long cr = wx * vy - wy * vx;
            d256 = isqrt64(((cr * cr) << 16) / L2);
        }
        const long long cov = min(256ll, max(0ll, (long long)thickness * 128 + 128 - d256));
        if (cov) {
            r += (int)((((long long)(col & 255u) - r) * cov + 128) >> 8);
            g += (int)((((long long)((col >> 8) & 255u) - g) * cov + 128) >> 8);
            b += (int)((((long long)((col >> 16) & 255u) - b) * cov + 128) >> 8);
        }
    }
    // np.flipud (:155): image row y is stored as row H - 1 - y
    out[((size_t)img * H + (H - 1 - y)) * W + x] = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
}

}  // namespace

extern "C" int salve_layout_rasterise(const salve_layout_t* layouts, int32_t n, const int32_t* poly_xy, const int32_t* segs, int32_t img_h,
                                      int32_t img_w, uint32_t* out, void* stream) {
    if (n == 0) return SALVE_OK;
    if (n < 0 || n > 65535 || !layouts || !out || img_h <= 0 || img_w <= 0) {
        salve_fail("salve_layout_rasterise: null pointer, bad size or more than 65535 images");
        return SALVE_ERR_BAD_ARG;
    }
    dim3 grid((unsigned)(((long long)img_h * img_w + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(layout_kernel, grid, dim3(256), 0, (hipStream_t)stream, layouts, poly_xy, segs, img_h, img_w, out);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}
