// layout.hip -- rasterised-LAYOUT modality for gfx950: filled room polygon + thick anti-aliased W/D/O segments per image.
//
// Stands behind salve/utils/bev_rendering_utils.py:104-251 (rasterize_single_layout: cv2.fillPoly :159-179, cv2.line with
// LINE_AA :210-251; np.flipud :155).  All pixel arithmetic is integer and follows oracle/layout_oracle.py rule for rule (the
// OpenCV arithmetic itself is "parity unpinned": cv2 is not in the image -- see that file's header):
//   * polygon: even-odd scanline fill on 16.16 fixed-point edge crossings (an edge covers y0 <= y < y1, a span runs from
//     ceil(left) to floor(right)) united with the 8-connected LineIterator pixels of every edge, in closed form;
//   * segment: OpenCV's ThickLine with LINE_AA -- anti-aliased quadrilateral + two anti-aliased 12-gon end caps, LineAA's
//     filter tables, the blend applied twice (round 3; round 2 used a capsule-distance rule of its own) -- see below.
// One thread per pixel, a 16 x 16 tile per workgroup; the primitives of a chunk of segments are set up in LDS by the tile's own
// threads.  The work is tiny next to the texture rasteriser (one launch renders thousands of layouts).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/salve_hip.h"
#include "salve_common.h"

namespace {

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

// ------------------------------------------------------------------------------------------------ cv2.line(LINE_AA), thick
// OpenCV 4.x drawing.cpp, as restated in oracle/layout_oracle.py (parity unpinned: cv2 is not in the image).  ThickLine =
// FillConvexPoly(quadrilateral) + an EllipseEx end cap (12-gon) at either end; FillConvexPoly = LineAA along every edge, then
// spans ceil(left) .. floor(right) filled with the colour; LineAA = three weighted pixels per major-axis step, blended twice.
// The oracle runs these routines as OpenCV does, as loops that write pixels.  Here a workgroup owns a 16 x 16 pixel tile:
// some of its threads SET UP the primitives of a chunk of segments in LDS (geometry in double precision exactly as drawing.cpp
// computes it, one thread per LineAA, one thread per polygon scan), then every thread evaluates its own pixel against the
// primitives in drawing order -- LineAA in closed form (step k = pixel's major coordinate - start).
constexpr int XY_SHIFT = 16;
constexpr long long XY_ONE = 1ll << XY_SHIFT;
constexpr int TILE = 16;
constexpr int SEGC = 7;                 // segments per chunk: 1 + 30 + 3 set-up tasks each
constexpr int CAP_N = 13;               // ellipse2Poly at 30-degree steps: 0, 30, ..., 360
constexpr int LINES_PER_SEG = 4 + 2 * CAP_N;

__constant__ unsigned char SLOPE_CORR[33] = {181, 181, 181, 182, 182, 183, 184, 185, 187, 188, 190, 192, 194, 194, 197, 198,
                                             201, 203, 206, 209, 211, 214, 218, 221, 224, 227, 231, 235, 238, 242, 246, 250, 254};
__constant__ unsigned char FILTER_TAB[64] = {168, 177, 185, 194, 202, 210, 218, 224, 231, 236, 241, 246, 249, 252, 254, 254,
                                             254, 254, 252, 249, 246, 241, 236, 231, 224, 218, 210, 202, 194, 185, 177, 168,
                                             158, 149, 140, 131, 122, 114, 105, 99, 91, 86, 79, 73, 67, 62, 57, 53,
                                             48, 44, 40, 37, 33, 30, 28, 25, 23, 21, 19, 17, 15, 14, 12, 11};
// SinTable (float, per degree) at 0, 30, ..., 450 degrees
__constant__ float SIN30[16] = {0.0f, 0.5f, 0.8660254f, 1.0f, 0.8660254f, 0.5f, 0.0f, -0.5f, -0.8660254f, -1.0f, -0.8660254f, -0.5f, -0.0f,
                                0.5f, 0.8660254f, 1.0f};

struct LineRec {       // one LineAA after its set-up
    long long v0, step;          // minor coordinate (16.16) at step 0 and its increment per major step
    int start, ecount;           // major pixel coordinate of step 0, last step
    short ep[9];
    short valid, xmajor;
};
struct SegRec {
    long long pts[LINES_PER_SEG][2];    // quadrilateral (4), cap at p0 (<= 13), cap at p1 (<= 13): 16.16 fixed point
    LineRec lines[LINES_PER_SEG];
    short span[3][TILE][2];             // per polygon and tile row: first / last filled pixel (first > last: nothing)
    int npts[3];                        // vertices per polygon (0: polygon absent)
    int active;                         // the thick line can touch this tile
    int thin;                           // thickness <= 1: ThickLine draws ONE LineAA from p0 to p1 (lines[0]) and nothing else
    unsigned colour;
};

__device__ __forceinline__ long long cv_round(double v) { return (long long)rint(v); }
__device__ __forceinline__ long long cdiv(long long a, long long b) { return a / b; }   // C division truncates toward zero

// clipLine(Size2l, Point2l&, Point2l&) of drawing.cpp
__device__ bool clip_line_fixed(long long w, long long h, long long& x1, long long& y1, long long& x2, long long& y2) {
    const long long right = w - 1, bottom = h - 1;
    if (w <= 0 || h <= 0) return false;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// LineAA's set-up (everything in front of its pixel loop)
__device__ void line_setup(LineRec& L, long long x1, long long y1, long long x2, long long y2, int W, int H) {
    L.valid = 0;
    if (!clip_line_fixed((long long)W << XY_SHIFT, (long long)H << XY_SHIFT, x1, y1, x2, y2)) return;
    long long dx = x2 - x1, dy = y2 - y1;
    const long long j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
    const long long i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
    long long step, fi, fj;
    int slope;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        if (j) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }   // the XOR swap under the mask j
        step = cdiv(dy << XY_SHIFT, ax | 1);
        x2 += XY_ONE;
        L.ecount = (int)((x2 >> XY_SHIFT) - (x1 >> XY_SHIFT));
        const long long jj = -(x1 & (XY_ONE - 1));
        y1 += ((step * jj) >> XY_SHIFT) + (XY_ONE >> 1);
        slope = (int)((step >> (XY_SHIFT - 5)) & 0x3f);
        slope ^= step < 0 ? 0x3f : 0;
        fi = (x1 >> (XY_SHIFT - 7)) & 0x78;
        fj = (x2 >> (XY_SHIFT - 7)) & 0x78;
        L.xmajor = 1; L.start = (int)(x1 >> XY_SHIFT); L.v0 = y1;
    } else {
        dx = (dx ^ i) - i;
        if (i) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
        step = cdiv(dx << XY_SHIFT, ay | 1);
        y2 += XY_ONE;
        L.ecount = (int)((y2 >> XY_SHIFT) - (y1 >> XY_SHIFT));
        const long long jj = -(y1 & (XY_ONE - 1));
        x1 += ((step * jj) >> XY_SHIFT) + (XY_ONE >> 1);
        slope = (int)((step >> (XY_SHIFT - 5)) & 0x3f);
        slope ^= step < 0 ? 0x3f : 0;
        fi = (y1 >> (XY_SHIFT - 7)) & 0x78;
        fj = (y2 >> (XY_SHIFT - 7)) & 0x78;
        L.xmajor = 0; L.start = (int)(y1 >> XY_SHIFT); L.v0 = x1;
    }
    L.step = step;
    slope = (slope & 0x20) ? 0x100 : SLOPE_CORR[slope];
    const int t0 = slope << 7, t1 = (int)((0x78 - fi) | 4) * slope, t2 = (int)(fj | 4) * slope;
    L.ep[0] = 0;
    L.ep[8] = (short)slope;
    L.ep[1] = L.ep[3] = (short)((((int)(((fj - fi) & 0x78) | 4) * slope) >> 8) & 0x1ff);
    L.ep[2] = (short)((t1 >> 8) & 0x1ff);
    L.ep[4] = (short)((((int)(((fj - fi) + 0x80) | 4) * slope) >> 8) & 0x1ff);
    L.ep[5] = (short)(((t1 + t0) >> 8) & 0x1ff);
    L.ep[6] = (short)((t2 >> 8) & 0x1ff);
    L.ep[7] = (short)(((t2 + t0) >> 8) & 0x1ff);
    L.valid = 1;
}

// One pixel against one LineAA: step k = the pixel's major coordinate - start; the three pixels of the step lie at minor
// coordinates (v >> 16) - 1 .. + 1 with weights FilterTable[dist + 32], [dist], [63 - dist]; ICV_PUT_POINT blends twice.
__device__ __forceinline__ void line_apply(const LineRec& L, int px, int py, unsigned colour, int& r, int& g, int& b) {
    if (!L.valid) return;
    const int k = (L.xmajor ? px : py) - L.start;
    if (k < 0 || k > L.ecount) return;
    const long long v = L.v0 + (long long)k * L.step;
    const int rr = (L.xmajor ? py : px) - ((int)(v >> XY_SHIFT) - 1);
    if (rr < 0 || rr > 2) return;
    const int ec = L.ecount - k;
    const int e = L.ep[(((k >= 2) + 1) & (k | 2)) * 3 + (((ec >= 2) + 1) & (ec | 2))];
    const int dist = (int)(v >> (XY_SHIFT - 5)) & 31;
    const int f = FILTER_TAB[rr == 0 ? dist + 32 : (rr == 1 ? dist : 63 - dist)];
    const int a = ((e * f) >> 8) & 0xff;
    const int cr = (int)(colour & 255u), cg = (int)((colour >> 8) & 255u), cb = (int)((colour >> 16) & 255u);
    r += ((cr - r) * a + 127) >> 8; r += ((cr - r) * a + 127) >> 8;
    g += ((cg - g) * a + 127) >> 8; g += ((cg - g) * a + 127) >> 8;
    b += ((cb - b) * a + 127) >> 8; b += ((cb - b) * a + 127) >> 8;
}

// FillConvexPoly's scan (LINE_AA, shift = XY_SHIFT) run by ONE thread: the spans of the rows ty0 .. ty0 + 15 of this tile.
__device__ void poly_scan(const long long (*v)[2], int npts, int W, int H, int ty0, short (*span)[2]) {
    for (int i = 0; i < TILE; i++) { span[i][0] = 1; span[i][1] = 0; }
    if (npts < 3) return;
    const long long delta = XY_ONE >> 1, delta1 = XY_ONE - 1, delta2 = 0;
    long long xmin = v[0][0], xmax = v[0][0], ymin = v[0][1], ymax = v[0][1];
    int imin = 0;
    for (int i = 0; i < npts; i++) {
        if (v[i][1] < ymin) { ymin = v[i][1]; imin = i; }
        ymax = max(ymax, v[i][1]); xmax = max(xmax, v[i][0]); xmin = min(xmin, v[i][0]);
    }
    xmin = (xmin + delta) >> XY_SHIFT; xmax = (xmax + delta) >> XY_SHIFT;
    ymin = (ymin + delta) >> XY_SHIFT; ymax = (ymax + delta) >> XY_SHIFT;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= W || (int)ymin >= H) return;
    ymax = min(ymax, (long long)H - 1);
    if ((int)ymin > ty0 + TILE - 1) return;   // the polygon starts below this tile
    int e_idx[2] = {imin, imin}, e_di[2] = {1, npts - 1}, e_ye[2] = {(int)ymin, (int)ymin};
    long long e_x[2] = {-XY_ONE, -XY_ONE}, e_dx[2] = {0, 0};
    int edges = npts;
    int y = (int)ymin;
    const int y_stop = min((int)ymax, ty0 + TILE - 1);   // rows below the tile do not matter to it
    do {
        if (y < (int)ymax || y == (int)ymin) {
            for (int i = 0; i < 2; i++) {
                if (y >= e_ye[i]) {
                    int idx0 = e_idx[i];
                    const int di = e_di[i];
                    int idx = idx0 + di;
                    if (idx >= npts) idx -= npts;
                    for (; edges-- > 0;) {
                        const int ty = (int)((v[idx][1] + delta) >> XY_SHIFT);
                        if (ty > y) {
                            const long long xs = v[idx0][0], xe = v[idx][0];
                            e_ye[i] = ty;
                            e_dx[i] = cdiv((xe - xs) * 2 + (ty - y), 2ll * (ty - y));
                            e_x[i] = xs;
                            e_idx[i] = idx;
                            break;
                        }
                        idx0 = idx;
                        idx += di;
                        if (idx >= npts) idx -= npts;
                    }
                }
            }
        }
        if (edges < 0) break;
        if (y >= ty0 && y < ty0 + TILE) {   // (ty0 >= 0)
            const int left = e_x[0] > e_x[1] ? 1 : 0, right = 1 - left;
            int xx1 = (int)((e_x[left] + delta1) >> XY_SHIFT);
            int xx2 = (int)((e_x[right] + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < W) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= W) xx2 = W - 1;
                span[y - ty0][0] = (short)xx1;
                span[y - ty0][1] = (short)xx2;
            }
        }
        e_x[0] += e_dx[0];
        e_x[1] += e_dx[1];
    } while (++y <= y_stop);
}

// ThickLine's geometry: quadrilateral, the two end-cap polygons (EllipseEx -> ellipse2Poly, rounded to fixed point)
__device__ void thick_line_setup(SegRec& S, int x1, int y1, int x2, int y2, int thickness) {
    const long long p0x = (long long)x1 << XY_SHIFT, p0y = (long long)y1 << XY_SHIFT, p1x = (long long)x2 << XY_SHIFT, p1y = (long long)y2 << XY_SHIFT;
    const double inv = 1.0 / (double)XY_ONE;
    const double dx = (double)(p0x - p1x) * inv, dy = (double)(p1y - p0y) * inv;
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    const long long th = (long long)thickness << (XY_SHIFT - 1);
    S.npts[0] = 0;
    if (fabs(r) > 2.220446049250313e-16) {
        r = ((double)th + odd * (double)XY_ONE * 0.5) / sqrt(r);
        const long long dpx = cv_round(dy * r), dpy = cv_round(dx * r);
        S.pts[0][0] = p0x + dpx; S.pts[0][1] = p0y + dpy;
        S.pts[1][0] = p0x - dpx; S.pts[1][1] = p0y - dpy;
        S.pts[2][0] = p1x - dpx; S.pts[2][1] = p1y - dpy;
        S.pts[3][0] = p1x + dpx; S.pts[3][1] = p1y + dpy;
        S.npts[0] = 4;
    }
    for (int c = 0; c < 2; c++) {
        const long long cx = c ? p1x : p0x, cy = c ? p1y : p0y;
        long long (*out)[2] = S.pts + 4 + CAP_N * c;
        const float alpha = SIN30[15], beta = SIN30[0];   // sincos(0): cos, sin
        int n = 0;
        long long prevx = -1, prevy = -1;
        bool have = false;
        // EllipseEx: the polygon's step from the axis length (90 degrees below 3 px, 30 below 10; larger radii -- 18 and 5 degrees --
        // do not occur: the reference draws 8- and 2-pixel lines, bevparams.py:81-99, and the host refuses thicker ones)
        const int dstep = (int)((th + (XY_ONE >> 1)) >> XY_SHIFT) < 3 ? 90 : 30;
        for (int a = 0; a <= 360; a += dstep) {
            const double x = (double)th * (double)SIN30[(450 - a) / 30], y = (double)th * (double)SIN30[a / 30];
            const double px = (double)cx + x * (double)alpha - y * (double)beta;
            const double py = (double)cy + x * (double)beta + y * (double)alpha;
            long long qx = cv_round(px / (double)XY_ONE) << XY_SHIFT, qy = cv_round(py / (double)XY_ONE) << XY_SHIFT;
            qx += cv_round(px - (double)qx);
            qy += cv_round(py - (double)qy);
            if (!have || qx != prevx || qy != prevy) { out[n][0] = qx; out[n][1] = qy; n++; prevx = qx; prevy = qy; have = true; }
        }
        if (n <= 1) { out[0][0] = out[1][0] = cx; out[0][1] = out[1][1] = cy; n = 2; }
        S.npts[1 + c] = n;
    }
}

__global__ __launch_bounds__(256) void layout_kernel(const salve_layout_t* __restrict__ layouts, const int32_t* __restrict__ poly_xy,
                                                     const int32_t* __restrict__ segs, int H, int W, uint32_t* __restrict__ out,
                                                     int32_t* __restrict__ status) {
    __shared__ SegRec seg[SEGC];
    const int img = blockIdx.y;
    const int tiles_x = (W + TILE - 1) / TILE;
    const int tx0 = (blockIdx.x % tiles_x) * TILE, ty0 = (blockIdx.x / tiles_x) * TILE;
    const int tid = threadIdx.x;
    const int x = tx0 + (tid & (TILE - 1)), y = ty0 + (tid >> 4);
    const bool in_img = x < W && y < H;
    const salve_layout_t L = layouts[img];
    const int32_t* P = poly_xy + 2 * (size_t)L.poly_off;
    int r = 0, g = 0, b = 0;
    if (in_img && L.n_poly > 0) {
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
    const int32_t* Sg = segs + 8 * (size_t)L.seg_off;
    for (int s0 = 0; s0 < L.n_seg; s0 += SEGC) {
        const int ns = min(SEGC, L.n_seg - s0);
        __syncthreads();   // the previous chunk's records are no longer read
        if (tid < ns) {    // stage A: geometry; can this thick line touch the tile at all?
            const int32_t* q = Sg + 8 * (size_t)(s0 + tid);
            SegRec& S = seg[tid];
            const int thickness = q[5];
            S.colour = (uint32_t)q[4];
            const int reach = thickness / 2 + 3;
            S.active = !(max(q[0], q[2]) + reach < tx0 || min(q[0], q[2]) - reach > tx0 + TILE - 1 ||
                         max(q[1], q[3]) + reach < ty0 || min(q[1], q[3]) - reach > ty0 + TILE - 1);
            // End caps of radius >= 10 pixels (thickness >= 19) are 20- and 72-gons in OpenCV (EllipseEx: 18- and 5-degree steps), which
            // this kernel does not draw: such a segment is NOT drawn at all and the launch says so -- never a silent 12-gon.
            if (((((long long)thickness << (XY_SHIFT - 1)) + (XY_ONE >> 1)) >> XY_SHIFT) >= 10) {
                S.active = false;
                if (status && blockIdx.x == 0) atomicOr(status, SALVE_STATUS_LAYOUT_THICKNESS);
            }
            S.thin = thickness <= 1;
            if (S.active && !S.thin) thick_line_setup(S, q[0], q[1], q[2], q[3], thickness);
            if (S.active && S.thin) {
                S.npts[0] = S.npts[1] = S.npts[2] = 0;
                S.pts[0][0] = (long long)q[0] << XY_SHIFT; S.pts[0][1] = (long long)q[1] << XY_SHIFT;
                S.pts[1][0] = (long long)q[2] << XY_SHIFT; S.pts[1][1] = (long long)q[3] << XY_SHIFT;
            }
        }
        __syncthreads();
        if (tid < ns * (LINES_PER_SEG + 3)) {   // stage B: one thread per LineAA, one per polygon scan
            const int si = tid / (LINES_PER_SEG + 3), j = tid % (LINES_PER_SEG + 3);
            SegRec& S = seg[si];
            if (S.active && S.thin) {
                if (j == 0) line_setup(S.lines[0], S.pts[0][0], S.pts[0][1], S.pts[1][0], S.pts[1][1], W, H);
            } else if (S.active) {
                if (j < LINES_PER_SEG) {
                    const int poly = j < 4 ? 0 : (j < 4 + CAP_N ? 1 : 2);
                    const int base = poly == 0 ? 0 : (poly == 1 ? 4 : 4 + CAP_N), i = j - base, n = S.npts[poly];
                    S.lines[j].valid = 0;
                    if (i < n) {   // FillConvexPoly: LineAA(v[i - 1], v[i]), starting from v[npts - 1]
                        const int ip = i == 0 ? n - 1 : i - 1;
                        line_setup(S.lines[j], S.pts[base + ip][0], S.pts[base + ip][1], S.pts[base + i][0], S.pts[base + i][1], W, H);
                    }
                } else {
                    const int poly = j - LINES_PER_SEG;
                    const int base = poly == 0 ? 0 : (poly == 1 ? 4 : 4 + CAP_N);
                    poly_scan(S.pts + base, S.npts[poly], W, H, ty0, S.span[poly]);
                }
            }
        }
        __syncthreads();
        if (in_img) {   // stage C: this pixel against the chunk's primitives, in drawing order
            const int row = y - ty0;
            for (int si = 0; si < ns; si++) {
                const SegRec& S = seg[si];
                if (!S.active) continue;
                const uint32_t col = S.colour;
                if (S.thin) { line_apply(S.lines[0], x, y, col, r, g, b); continue; }
#pragma unroll 1
                for (int poly = 0; poly < 3; poly++) {
                    const int base = poly == 0 ? 0 : (poly == 1 ? 4 : 4 + CAP_N), n = S.npts[poly];
                    if (n < 1) continue;
                    for (int i = 0; i < n; i++) line_apply(S.lines[base + i], x, y, col, r, g, b);
                    if (x >= S.span[poly][row][0] && x <= S.span[poly][row][1]) { r = (int)(col & 255u); g = (int)((col >> 8) & 255u); b = (int)((col >> 16) & 255u); }
                }
            }
        }
    }
    // np.flipud (:155): image row y is stored as row H - 1 - y
    if (in_img) out[((size_t)img * H + (H - 1 - y)) * W + x] = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
}

}  // namespace

extern "C" int salve_layout_rasterise(const salve_layout_t* layouts, int32_t n, const int32_t* poly_xy, const int32_t* segs, int32_t img_h,
                                      int32_t img_w, uint32_t* out, int32_t* status, void* stream) {
    if (n == 0) return SALVE_OK;
    if (n < 0 || n > 65535 || !layouts || !out || img_h <= 0 || img_w <= 0 || img_h > 32000 || img_w > 32000) {
        salve_fail("salve_layout_rasterise: null pointer, bad size or more than 65535 images");
        return SALVE_ERR_BAD_ARG;
    }
    dim3 grid((unsigned)(((img_w + TILE - 1) / TILE) * ((img_h + TILE - 1) / TILE)), (unsigned)n);
    hipLaunchKernelGGL(layout_kernel, grid, dim3(256), 0, (hipStream_t)stream, layouts, poly_xy, segs, img_h, img_w, out, status);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}
