// bev_render.hip -- BEV texture-map rasteriser for gfx950 (MI355X).  Compile with -ffp-contract=off.
//
// One render = one panorama surface under one Sim(2) pose -> one 501x501 BEV image.  Kernels:
//
//   bev_pano_index_kernel / bev_splat_kernel   (bev_splat.h; round 4)  pose-independent block boxes per (panorama, surface), then per
//                        (render, 128x128 output tile): cull blocks by their posed boxes, back-project, pose, prune, round to the
//                        BEV pixel, z-order by ds_max_u32 on a key tile in LDS, emit the sparse image tile + bitmap words.
//                        Reference: bev_rendering_utils.py:367-413 (back-projection), :443-451 (pose),
//                        :274-287 (prune + pixel index), zorder_utils.py:10-83 (winner), :307-308 (sparse image).
//   bev_densify_kernel   (LDS/VALU)  one workgroup per render: occupancy + "non-empty" bitmaps from the splat's words into LDS
//                        (2 x 32 KB), 11x11
//                        dilation mask on the bitmaps, then every site walks its own Delaunay star (star_local.h,
//                        star_delaunay.h) and the owned triangles are rasterised with exact rational barycentric
//                        weights into the output image.  Reference: interpolation_utils.py:21-54 (griddata linear), :74-122 (mask),
//                        bev_rendering_utils.py:318-319 (mask multiply, flipud).
//   bev_tile_kernel      BEV -> verifier input tile (resize 234, crop 224, normalise).  Reference:
//                        train_utils.py:126-159, transform.py:256-272, 386-420, 105-123, 177-202.
//
// Exactness: pixel indices follow the reference's float64 op order bit for bit (float32 depth product, float64
// table products, the FMA order of numpy's BLAS matmul, half-to-even rounding); everything after the pixel index
// is integer arithmetic.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/salve_hip.h"
#if defined(SALVE_PROFILE_WALK)
// development build: count the general walk's work in LDS (read back through dbg_stats slots 1, 2, 3)
enum { SDC_apex = 0, SDC_apex_slow, SDC_apex_far, SDC_rows, SDC_bits, SDC_exact, SDC_apex_table, SDC_apex_cached, SDC_N };
__shared__ int sd_counters[SDC_N];
#if defined(SALVE_PROFILE_NO_COUNT)   // timers only: the counters' LDS atomics (one per row and per bit) distort the laps
#define SD_COUNT(c)
#else
#define SD_COUNT(c) atomicAdd(&sd_counters[SDC_##c], 1)
#endif
// ... and where its time goes: wave-clock laps (units of 16 cycles), lane 0 of each wave
enum { SDP_table = 0, SDP_window, SDP_share, SDP_slow, SDP_far, SDP_e2_total, SDP_e1_total, SDP_nearest, SDP_N };
__shared__ int sd_timers[SDP_N];
// ... and the kernel's phases (dbg_flags & 128): [0] up to the site list and the mask [1] E1 [2] E2 [3] F [4] G
__shared__ int sd_phase[8];
#define SD_HARD_REASON(slot) atomicAdd(&sd_phase[slot], 1)   // [6] edge beyond the table's range [7] table exhausted
#define SD_PHASE(slot, t) { const long long now_ = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&sd_phase[slot], (int)((now_ - (t)) >> 4)); (t) = now_; }
#define SD_NOW() clock64()
#define SD_LAP(slot, t) { const long long now_ = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&sd_timers[SDP_##slot], (int)((now_ - (t)) >> 4)); (t) = now_; }
#endif
#include "star_delaunay.h"
#include "star_local.h"
#include "star_table.h"
#include <mutex>
#include "salve_common.h"

namespace {

constexpr int DENSIFY_THREADS = 512;
constexpr int MASK_ROWS_PER_TASK = 16;
constexpr int MASK_MAX_HALF = 8;
constexpr int N_SCAL = 24;   // int32 scalars of a render in LDS (bev_densify_kernel: scal[])
#ifndef SALVE_E2_GROUP
#define SALVE_E2_GROUP 64
#endif
#ifndef SALVE_HARD_RUN
#define SALVE_HARD_RUN 8
#endif
constexpr int E2_GROUP = SALVE_E2_GROUP;   // lanes that share the general walk of one hard site (power of two, <= 64)
constexpr int HARD_RUN = SALVE_HARD_RUN;   // consecutive hard-list entries a group takes at a time
// z-order key of a pixel: (unit slice + 1) << 21 | point index.  atomicMax keeps the highest slice and, within it, the
// last point in raster order (zorder_utils.py:10-83 + "last index wins" of the sparse image); 0 = no point.  The
// colour is NOT in the key: the densify kernel fetches it from the point's source array by index.
constexpr int KEY_SLICE_SHIFT = 21;
constexpr uint32_t KEY_INDEX_MASK = (1u << KEY_SLICE_SHIFT) - 1u;

struct DevCfg {
    int pano_h, pano_w, crop_rows, rows, npts;
    int H, W, wpr, mask_half;
    float depth_scale;
    double xmin, xmax, ymin, ymax, tx, ty, scale;
    double rp00, rp01, rp10, rp11;
    double zlo[2], zhi[2], zmin;
    int nslices;
    int out_flags;  // 1: do not flip the image vertically, 2: no hallucination mask (plain interpolation), 4: densify in the given order
    int dbg_flags;  // development only: 1 = skip the star phase, 2 = walk stars but do not rasterise
};

// What the densify kernel needs of the configuration: six integers (the full DevCfg with its 19 doubles, passed by
// value, cost the kernel 101 SGPR spills).
struct DensifyCfg {
    int H, W, wpr, mask_half, out_flags, dbg_flags;
};

#include "bev_splat.h"

// ------------------------------------------------------------------------------------------------ utility scatters
// Splat of an explicit coloured point cloud (the `xyzrgb` argument of render_bev_image): world-frame points, no
// back-projection and no pose; prune -> pixel index -> z-order key into a key image in memory (one render, a utility path:
// bev_emit_keys_kernel turns it into the sparse image and the bitmaps).  The point's position in the list is its index.
__global__ __launch_bounds__(256) void bev_scatter_points_kernel(DevCfg c, const double* __restrict__ xyz, int npts,
                                                                 uint32_t* __restrict__ kimg, int* __restrict__ n_in_window) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    const double x1 = xyz[3 * (size_t)i], y1 = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    if (!(c.xmin <= x1 && x1 <= c.xmax && c.ymin <= y1 && y1 <= c.ymax)) return;
    atomicAdd(n_in_window, 1);
    const int ix = (int)rint((x1 + c.tx) * c.scale);   // (identity product dropped, as in bev_splat_kernel)
    const int iy = (int)rint((y1 + c.ty) * c.scale);
    const double zs = floor(z) - c.zmin;
    if (!(zs >= 0.0 && zs < (double)c.nslices) || ix < 0 || ix >= c.W || iy < 0 || iy >= c.H) return;
    atomicMax(kimg + (size_t)iy * c.W + ix, ((uint32_t)((int)zs + 1) << KEY_SLICE_SHIFT) | (uint32_t)i);
}

// ------------------------------------------------------------------------------------------------ densify
struct RasterEmit {
    int H, W, wpr;
    const uint32_t* occ;
    const uint32_t* msk;
    uint32_t* bev;  // output image of this render; the data pixels (triangle vertices) already hold their colours
    int flip;       // H - 1 to flip vertically (np.flipud), -1: no flip
    int lane, nlanes;  // rows lane, lane + nlanes, ... of the bounding box (several lanes may share one triangle)
    bool skip;
    int32_t* status;   // device status word: a vertex outside the image is REPORTED and the triangle left out (fence, below)
    // pixel (x, y) of the output image as a 32-bit element index (the image is smaller than 2048 x 2048; a 64-bit index product is
    // a quarter-rate v_mad_u64_u32 per access)
    __device__ __forceinline__ uint32_t pix(int x, int y) const { return (uint32_t)(((flip >= 0 ? flip - y : y) * W) + x); }

    // floor(n / d) for d > 0, |n| < 2^23, with inv_d ~ 1 / d: the float quotient is within 1 of the answer (n is exact in float32,
    // the product is off by 2^-22 relative at most), and the remainder says which way.  The GPU
    // has no integer divide: the compiler's expansion is ~35 instructions, and the span of a wide triangle needs three per row.
    static __device__ __forceinline__ int floor_div(int n, int d, float inv_d) {
        int q = (int)floorf((float)n * inv_d);
        const int r = n - q * d;
        q += (r >= d) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    }

    // floor((wa ca + wb cb + wc cc) / area) per colour byte, exactly: weights < area < 2^23 (coordinates < 2048), colours < 256, so
    // the numerator is below 2^31 and 24-bit multiply-adds compute it exactly; the quotient (<= 255) comes from a float product
    // that is within 6e-5 of it, fixed up with the exact remainder.  (Until round 3: int64 products and a float64 division --
    // ~150 cycles per pixel where this takes ~40.)
    static __device__ __forceinline__ uint32_t blend(int32_t wa, int32_t wb, int32_t wc, int32_t area, float inv_area, uint32_t ca,
                                                     uint32_t cb, uint32_t cc) {
        uint32_t out = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const uint32_t num = __umul24((uint32_t)wa, (ca >> (8 * ch)) & 255u) + __umul24((uint32_t)wb, (cb >> (8 * ch)) & 255u) +
                                 __umul24((uint32_t)wc, (cc >> (8 * ch)) & 255u);
            uint32_t q = (uint32_t)((float)num * inv_area);
            const int32_t r = (int32_t)(num - __umul24(q, (uint32_t)area));
            q += (r >= area) ? 1u : 0u;
            q -= (r < 0) ? 1u : 0u;
            out |= q << (8 * ch);
        }
        return out;
    }

    __device__ __forceinline__ void operator()(int ax, int ay, int bx, int by, int cx, int cy) const {
        const int32_t area = sd_orient(ax, ay, bx, by, cx, cy);
        // area == 1 (twice the area): by Pick's theorem the triangle holds no lattice point besides its vertices
        if (area <= 1 || skip) return;
        int x0 = min(ax, min(bx, cx)), x1 = max(ax, max(bx, cx));
        const int y0 = min(ay, min(by, cy)), y1 = max(ay, max(by, cy));
        // Fence: the rows and words below are addressed from the vertices without further tests, so a walk that
        // handed over anything but three sites of THIS image -- the SLP-vectorised build of the general walk did, DESIGN.md section 8 --
        // would read and write outside the render's buffers.  Such a triangle is left out and the render reported as failed: a wrong
        // image that names itself instead of a memory fault.  (Three vector instructions per triangle of the general walk.)
        if ((x0 | y0) < 0 || x1 >= W || y1 >= H) {
            if (status && lane == 0) atomicOr(status, SALVE_STATUS_WALK_FAILED);
            return;
        }
        uint32_t ca = 0, cb = 0, cc = 0;
        bool have = false;
        const bool wide = (x1 - x0) > 40;
        const float inv_area = 1.0f / (float)area;
        // the three edge functions E_i(x, y) = A_i x + Bx_i (y - Py_i) + C_i >= 0 inside (counter-clockwise), for the exact
        // span of a wide triangle on a row
        const int ex[3][4] = {{bx, by, cx, cy}, {cx, cy, ax, ay}, {ax, ay, bx, by}};
        int eA[3], eB[3], eC[3];
        float einv[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int px = ex[e][0], py = ex[e][1], qx = ex[e][2], qy = ex[e][3];
            eA[e] = -(qy - py);
            eB[e] = qx - px;
            eC[e] = (qy - py) * px - (qx - px) * py;   // E(x, y) = eA x + eB y + eC
            einv[e] = eA[e] != 0 ? 1.0f / (float)(eA[e] < 0 ? -eA[e] : eA[e]) : 0.f;
        }
        for (int y = y0 + lane; y <= y1; y += nlanes) {
            int xa = x0, xb = x1;
            if (wide) {
                // exact span of the triangle on this row: intersect the three half-planes A_i x + D_i >= 0
                bool empty = false;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const int32_t A = eA[e];
                    const int32_t D = eB[e] * y + eC[e];
                    if (A > 0) {  // x >= ceil(-D / A)
                        xa = max(xa, floor_div(-D + A - 1, A, einv[e]));
                    } else if (A < 0) {  // x <= floor(D / -A)
                        xb = min(xb, floor_div(D, -A, einv[e]));
                    } else if (D < 0) {
                        empty = true;
                    }
                }
                if (empty || xa > xb) continue;
            }
            for (int w = xa >> 5; w <= (xb >> 5); w++) {
                uint32_t bits = msk[(y * wpr) + w] & ~occ[(y * wpr) + w];
                const int lo = xa - (w << 5), hi = xb - (w << 5);
                if (lo > 0) bits &= 0xFFFFFFFFu << lo;
                if (hi < 31) bits &= 0xFFFFFFFFu >> (31 - hi);
                while (bits) {
                    const int x = (w << 5) + (__ffs((int)bits) - 1);
                    bits &= bits - 1;
                    const int32_t wa = sd_orient(bx, by, cx, cy, x, y);
                    const int32_t wb = sd_orient(cx, cy, ax, ay, x, y);
                    const int32_t wc = area - wa - wb;
                    if ((wa | wb | wc) < 0) continue;
                    if (!have) {
                        // vertex colours from the output image, where phase B of this workgroup put them: read past
                        // the L1 (the stores went to L2; the L1 may hold older lines of these addresses)
                        ca = __hip_atomic_load(bev + pix(ax, ay), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cb = __hip_atomic_load(bev + pix(bx, by), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cc = __hip_atomic_load(bev + pix(cx, cy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        have = true;
                    }
                    bev[pix(x, y)] = blend(wa, wb, wc, area, inv_area, ca, cb, cc);
                }
            }
        }
    }
};

// Site list, hard-site list and triangle queues are written and read by ONE workgroup: plain stores (write-back in this
// XCD's L2), made visible to the workgroup's other waves by wg_barrier_after_global_stores(), read back past the L1.  As
// agent-scope stores they were written THROUGH to memory -- the L2s of the eight XCDs are not coherent with each other --,
// one 32-byte sector per 4- or 8-byte store whenever the lanes of an instruction did not fill sectors: 1.3 MB of HBM writes
// per render for a 0.2 MB site list (profiles/r02_pmc_traffic.md).
template <class T>
__device__ __forceinline__ void store_wb(T* p, T v) { *p = v; }

// Hard-site list: two words per entry, the site (y << 16 | x) and the state its lean walk was in when it gave up
// (sd_star_resume: a + 128 in two bytes, n0 + 1 in two bits each, bit 20 counter-clockwise, bit 21 half walk), or
// HARD_FRESH if there is nothing to take over.  A list that overflows (every second pixel a hard site) is a failed render.
constexpr uint32_t HARD_FRESH = 1u << 22;
__device__ __forceinline__ void push_hard(uint32_t* list, int* counter, int cap, uint32_t site, uint32_t state, int32_t* status) {
    const int k = atomicAdd(counter, 1);
    if (k < cap) {
        *reinterpret_cast<uint2*>(list + 2 * k) = make_uint2(site, state);
    } else if (status) {
        atomicOr(status, SALVE_STATUS_WALK_FAILED);
    }
}

// Emit functor of the local star walk: an owned triangle that has anything to fill (twice its area > 1: a unit lattice
// triangle holds no lattice point besides its vertices, Pick) goes to the render's queue, 8 bytes: the site and the two other
// vertices relative to it; all lanes rasterise the queue afterwards (phase F).
//
// ONE queue, ONE path here (round 3).  Rounds 1-2 sorted the triangles into three queues in this functor -- midpoint triangles
// (twice-the-area 2), centroid triangles (3, lattice centroid), general ones -- so that phase F could run each kind with a loop
// of its own; but the lean loop executes the sum of all paths any of its 64 lanes takes, and the sorting was a quarter of its
// instructions (161 k of 611 k per render).  Phase F sorts now, where an entry is seen once by one lane.
// The queue cannot overflow: every triangle in it holds a lattice point that is not a site (inside: in no other triangle; on
// an edge: in one other), so there are at most 2 (H W - n) of them, and at most 2 n triangles in all: <= H W, the queue's size.
struct QueueEmit {
    unsigned long long* queue;
    int* counter;  // LDS
    int capacity;  // H W
    bool drop;     // development (dbg_flags & 1024): timing only, nothing is queued
    int32_t* status;
    __device__ __forceinline__ void operator()(int ax, int ay, int bx, int by, int cx, int cy) const {
        if (drop) return;
        const int32_t area = sd_orient(ax, ay, bx, by, cx, cy);
        if (area <= 1) return;
        const int slot = atomicAdd(counter, 1);
        if (slot < capacity) {
            const uint32_t rel = (uint32_t)((bx - ax) & 0xFF) | ((uint32_t)((by - ay) & 0xFF) << 8) |
                                 ((uint32_t)((cx - ax) & 0xFF) << 16) | ((uint32_t)((cy - ay) & 0xFF) << 24);
            store_wb(queue + slot, ((unsigned long long)rel << 32) | ((uint32_t)ay << 16) | (uint32_t)ax);
        } else if (status) {
            atomicOr(status, SALVE_STATUS_WALK_FAILED);   // (unreachable by the bound above; never silently)
        }
    }
};

// Workgroup barrier that also orders this workgroup's GLOBAL stores against what other waves of the workgroup do
// afterwards to the same addresses (later stores, or sc1 loads served by L2).  A plain __syncthreads() does not wait
// for outstanding stores on gfx950 (workgroup-scope release needs no vmcnt wait when all waves share a CU), so two
// waves' stores to one pixel could land out of order -- observed as rare wrong pixels once another kernel shared
// the memory pipeline.
__device__ __forceinline__ void wg_barrier_after_global_stores() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- the verifier's input tile (salve_hip.h: salve_bev_tiles) -- shared by the tile kernels below and by the densify kernel's phase H
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }  // |values| < 3: no saturation needed
__device__ __forceinline__ void tile_pixel(const uint32_t* __restrict__ img, int W, const int4 cy, const int4 cx, const float* __restrict__ lut, float v[3]) {
    const uint32_t p00 = img[(size_t)cy.x * W + cx.x], p01 = img[(size_t)cy.x * W + cx.y];
    const uint32_t p10 = img[(size_t)cy.y * W + cx.x], p11 = img[(size_t)cy.y * W + cx.y];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const int a = (p00 >> (8 * ch)) & 255, b = (p01 >> (8 * ch)) & 255;
        const int cc = (p10 >> (8 * ch)) & 255, d = (p11 >> (8 * ch)) & 255;
        const int S0 = a * cx.z + b * cx.w;  // horizontal pass, x2048
        const int S1 = cc * cx.z + d * cx.w;
        int r = (((cy.z * (S0 >> 4)) >> 16) + ((cy.w * (S1 >> 4)) >> 16) + 2) >> 2;
        r = min(max(r, 0), 255);
        v[ch] = lut[ch * 256 + r];
    }
}


// (r6) salve_bev_densify_tiles: the arguments of the tile phase, read by bev_densify_kernel from the workspace (behind order[] in the key
// image) instead of as kernel arguments -- the kernel runs at the limit of the scalar register file, and every live scalar pair moves
// spill reloads into its hot loops (tools/measure/densify_spills.py).  Written by bev_tile_fuse_kernel in front of the densify launch.
struct TileFuse {
    const salve_tile_job_t* jobs_a;   // [n] per RENDER of the launch: destination sample and channel (bev_offset unused: the image is the render's own)
    const salve_tile_job_t* jobs_b;   // [n] per render: the pair's second image -- element offset into tiles_b -- and its channel
    const uint32_t* tiles_b;          // SALVE_TILE_U8X4 images
    const int32_t* coef_y;
    const int32_t* coef_x;
    const float* lut;
    uint16_t* out;
    int32_t resize, crop, out_c, reserved;
};
__global__ void bev_tile_fuse_kernel(TileFuse t, TileFuse* dst) { *dst = t; }

// Apex candidates of short edges (star_table.h): built once per process on the host with the exact predicates.
__device__ SdTable d_star_table;

static int ensure_star_table() {
    static std::mutex mu;
    static unsigned long long uploaded = 0;  // bit per device ordinal
    static SdTable host_table;
    static int built = 0;                    // 0 not yet, 1 ok, -1 failed
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    SALVE_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && ((uploaded >> dev) & 1ull)) return SALVE_OK;
    if (built == 0) built = sdt_build(&host_table) ? 1 : -1;
    if (built < 0) { salve_fail("star table construction could not be certified"); return SALVE_ERR_UNSUPPORTED; }
    SALVE_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_star_table), &host_table, sizeof(SdTable)));
    if (dev >= 0 && dev < 64) uploaded |= 1ull << dev;
    return SALVE_OK;
}

// ---- longest renders first (r5).  A launch of n renders is n / 512 rounds of resident workgroups, dispatched in the order of their
// ids; renders differ in cost, and the launch ends with a tail in which CUs run dry behind the last, arbitrary renders.  With the
// costly renders dispatched first the tail is made of cheap ones: densify -4 ... -5 % on every synthetic scene (box 14.04 -> 13.32 ms
// per 4096, cluttered 19.24 -> 18.43, noisy 20.12 -> 19.64 with the order made on the host; tools/probe/densify_order_probe.py), identical images -- renders are independent.
// The cost estimate is a count the splat makes from a tile's occupancy words while they are in its LDS (bev_splat.h: emit_tile): sites with two or more of their four
// neighbours missing (outline and isolated sites: the ones whose walks are long; the plain site or point count does not predict the
// cluttered / noisy scenes, and of five such counts this one was good on all three scenes).
// bev_order_kernel: ONE workgroup, counting sort by cost[] (summed by the splat's tiles with one atomic each), descending, into
// order[] (ties in any order: the atomics of the scatter decide, nothing observable depends on it).
// uint32 words of one render's two bitmaps ([occupancy | non-empty][tile][TILE_H rows][TILE_WORDS]) in the workspace: the ONE definition of that
// layout, used by carve_workspace (host) and by bev_densify_kernel to find order[] behind the bitmaps of its launch.
__host__ __device__ inline size_t bitmap_words_hw(int H, int W) {
    return (size_t)2 * ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H) * TILE_H * TILE_WORDS;
}
constexpr int DENSIFY_ORDERED = 256;      // DensifyCfg::out_flags, set by bev_stage only: order[] is valid
constexpr int DENSIFY_TILES = 512;        // ... a TileFuse block is valid: phase H writes the render's verifier tile
// the key image of a launch of n renders: int32 cost[n], int32 order[n], then (16-byte aligned) the TileFuse block
__host__ __device__ inline size_t tile_fuse_word(size_t n) { return (2 * n + 3) & ~(size_t)3; }
constexpr int ORDER_BINS = 1024;
constexpr int ORDER_MIN_RENDERS = 1025;   // more than two rounds of the 512 resident workgroups.  Measured (tools/probe/densify_order_threshold.py, costly
                                          // first against as given): 1536 ... 4096 renders -2.2 ... -6.9 % on the box and the noisy scene; at 640 / 768 /
                                          // 1024 renders -7 ... +4 % with either sign (1024 = exactly two rounds: +3.5 % box, config 5's launches +5 %)

__global__ __launch_bounds__(1024) void bev_order_kernel(const int32_t* __restrict__ cost, int n, int32_t* __restrict__ order) {
    static_assert(ORDER_BINS == 1024, "one bin per thread of the one workgroup");
    __shared__ int bins[ORDER_BINS];
    __shared__ int cmax;
    const int tid = threadIdx.x;
    bins[tid] = 0;
    if (tid == 0) cmax = 1;
    __syncthreads();
    int m = 1;
    for (int i = tid; i < n; i += 1024) m = max(m, cost[i]);
    atomicMax(&cmax, m);
    __syncthreads();
    const float scale = (float)(ORDER_BINS - 1) / (float)cmax;
    auto bin_of = [&](int c) { return ORDER_BINS - 1 - min(ORDER_BINS - 1, max(0, (int)((float)c * scale))); };   // costly first
    for (int i = tid; i < n; i += 1024) atomicAdd(&bins[bin_of(cost[i])], 1);
    __syncthreads();
    if (tid == 0) {   // exclusive scan of 1024 counters by one lane: a microsecond
        int acc = 0;
        for (int b = 0; b < ORDER_BINS; b++) { const int c = bins[b]; bins[b] = acc; acc += c; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&bins[bin_of(cost[i])], 1)] = i;
}

// DEV = false is the product kernel: the development outputs (mask image, work counters) and the development flags of the
// configuration (phases switched off for timing, tools/densify_*.py) are compiled OUT -- their pointers and tests kept a dozen
// scalar registers live through every loop of a kernel that runs at the limit of the scalar register file (100 spills), and
// where the spill reloads land decides 10 % of its vector instructions (round 3).  DEV = true is launched whenever a caller
// passes a debug buffer or a non-zero flag word.
template <bool DEV>
__global__ __launch_bounds__(DENSIFY_THREADS, 4) void bev_densify_kernel(
    DensifyCfg c, const uint32_t* __restrict__ bitmaps_all, uint32_t* __restrict__ bev_all,
    uint32_t* __restrict__ sitelist_all, uint32_t* __restrict__ hardlist_all, unsigned long long* __restrict__ triq_all,
    uint8_t* __restrict__ dbg_mask_arg, int32_t* __restrict__ dbg_stats_arg, int16_t* __restrict__ dbg_aux, int32_t* __restrict__ status) {
    uint8_t* const dbg_mask = DEV ? dbg_mask_arg : nullptr;
    int32_t* const dbg_stats = DEV ? dbg_stats_arg : nullptr;
    const int dbg_flags = DEV ? c.dbg_flags : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = c.H, W = c.W, wpr = c.wpr;
    uint32_t* occ = reinterpret_cast<uint32_t*>(smem);
    uint32_t* msk = occ + H * wpr;
    int16_t* rmin = reinterpret_cast<int16_t*>(msk + H * wpr);
    const int Hp = (H + 1) & ~1;  // keep the int32 scalars 4-byte aligned behind the two int16 arrays
    int16_t* rmax = rmin + Hp;
    int* scal = reinterpret_cast<int*>(rmax + Hp);  // [0] n_sites [1] min x [2] max x [3] rows [4] steps [5] err
    unsigned long long* tri_cache = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(scal + N_SCAL) + 7) & ~(uintptr_t)7);  // SD_CACHE_SIZE entries

    __shared__ int list_wave_total[DENSIFY_THREADS / 64];
    // (bev_order_kernel: the costly renders first.  The order array lies behind the bitmaps of the launch's renders -- the workspace's
    //  key image: [cost n][order n] -- and is named by a flag bit instead of a pointer argument: one more live scalar register pair moved
    //  eight spill reloads into the lean-walk loop, tools/measure/densify_spills.py)
    int rid = blockIdx.x;
    if (c.out_flags & DENSIFY_ORDERED) {
        // order[] = the second int32 array of the workspace's key image, which carve_workspace puts behind the bitmaps of the launch's n = gridDim.x
        // renders (ONE layout function for host and kernel: bitmap_words_hw).  A value outside the launch -- a stale or foreign workspace -- must not
        // become an address: the workgroup leaves and the launch reports the failure.
        rid = reinterpret_cast<const int32_t*>(bitmaps_all + (size_t)gridDim.x * bitmap_words_hw(c.H, c.W))[gridDim.x + blockIdx.x];
        if ((unsigned)rid >= gridDim.x) {
            if (threadIdx.x == 0 && status) atomicOr(status, SALVE_STATUS_WALK_FAILED);
            return;
        }
    }
    const int flip = (c.out_flags & 1) ? -1 : H - 1;
    uint32_t* bev = bev_all + (size_t)rid * H * W;
    auto pixi = [&](int x, int y) -> uint32_t { return (uint32_t)((flip >= 0 ? flip - y : y) * W + x); };   // pixel (x, y) of the output image, 32-bit element index (RasterEmit::pix)
    uint32_t* sitelist = sitelist_all + (size_t)rid * H * W;
    uint32_t* hardlist = hardlist_all + (size_t)rid * H * W;
    unsigned long long* triq = triq_all + (size_t)rid * H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = DENSIFY_THREADS >> 6;
    const int p = c.mask_half;

#if defined(SALVE_PROFILE_WALK)
    if (tid < SDC_N) sd_counters[tid] = 0;
    if (tid < SDP_N) sd_timers[tid] = 0;
    if (tid < 8) sd_phase[tid] = 0;
    __syncthreads();
    long long t_phase = SD_NOW();
#endif
    for (int i = tid; i < SD_CACHE_SIZE; i += DENSIFY_THREADS) tri_cache[i] = 0ull;
    if (tid < N_SCAL) scal[tid] = (tid == 1) ? W : (tid == 2 ? -1 : (tid == 12 ? H : (tid == 13 ? -1 : 0)));  // [12] min y [13] max y  // [6] site cursor [7] hard sites [8] queued triangles
    __syncthreads();

    // ---- phase B: the bitmaps the splat emitted (bev_splat.h: [occupancy | non-empty][tile][128 rows][4 words]; the sparse image
    //      is already in `bev`) into LDS; then, a thread per row: the row's extent and site count, and the horizontal dilation of
    //      the "non-empty" bits in place (`msk`; uint8 channel product wraps mod 256: interpolation_utils.py:95).
    {
        const int tiles_x = (W + TILE_W - 1) / TILE_W, ntiles = tiles_x * ((H + TILE_H - 1) / TILE_H);
        const uint4* bm_occ = reinterpret_cast<const uint4*>(bitmaps_all + (size_t)rid * 2 * ntiles * TILE_H * TILE_WORDS);
        const uint4* bm_ne = bm_occ + (size_t)ntiles * TILE_H;
        for (int i = tid; i < ntiles * TILE_H; i += DENSIFY_THREADS) {
            const int t = i / TILE_H, r = i % TILE_H;
            const int y = (t / tiles_x) * TILE_H + r, w0 = (t % tiles_x) * TILE_WORDS;
            if (y >= H) continue;
            const uint4 o = bm_occ[i], n = bm_ne[i];
            const uint32_t ow[4] = {o.x, o.y, o.z, o.w}, nw[4] = {n.x, n.y, n.z, n.w};
#pragma unroll
            for (int k = 0; k < TILE_WORDS; k++)
                if (w0 + k < wpr) { occ[y * wpr + w0 + k] = ow[k]; msk[y * wpr + w0 + k] = nw[k]; }
        }
        __syncthreads();
        int cnt = 0, lo_all = W, hi_all = -1, rows = 0, y_min = H, y_max = -1;
        for (int y = tid; y < H; y += DENSIFY_THREADS) {
            uint32_t prev = 0u, cur = msk[y * wpr];
            int lo = W, hi = -1;
            for (int w = 0; w < wpr; w++) {
                const uint32_t next = (w + 1 < wpr) ? msk[y * wpr + w + 1] : 0u;
                uint32_t dil = cur;
                for (int d = 1; d <= p; d++) dil |= (cur << d) | (prev >> (32 - d)) | (cur >> d) | (next << (32 - d));
                msk[y * wpr + w] = dil;
                const uint32_t ob = occ[y * wpr + w];
                if (ob) {
                    cnt += __popc(ob);
                    if (hi < 0) lo = (w << 5) + __ffs((int)ob) - 1;
                    hi = (w << 5) + 31 - __clz((int)ob);
                }
                prev = cur;
                cur = next;
            }
            rmin[y] = (int16_t)lo;
            rmax[y] = (int16_t)hi;
            if (hi >= 0) {
                lo_all = min(lo_all, lo); hi_all = max(hi_all, hi);
                y_min = min(y_min, y); y_max = max(y_max, y);
                rows++;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            cnt += __shfl_xor(cnt, off); rows += __shfl_xor(rows, off);
            lo_all = min(lo_all, __shfl_xor(lo_all, off)); hi_all = max(hi_all, __shfl_xor(hi_all, off));
            y_min = min(y_min, __shfl_xor(y_min, off)); y_max = max(y_max, __shfl_xor(y_max, off));
        }
        if (lane == 0 && rows) {
            atomicAdd(&scal[0], cnt); atomicAdd(&scal[3], rows);
            atomicMin(&scal[1], lo_all); atomicMax(&scal[2], hi_all);
            atomicMin(&scal[12], y_min); atomicMax(&scal[13], y_max);
        }
    }
    __syncthreads();

    // ---- phase B2: the list of the sites that need a walk, in raster order (star_local.h: a site between five site
    //      neighbours owns unit triangles only and is left out -- about half of all sites).  Every thread takes a run of
    //      consecutive bitmap words; an exclusive scan of the counts gives its place in the list.
    {
        const int total = H * wpr, run = (total + DENSIFY_THREADS - 1) / DENSIFY_THREADS;
        const int w_begin = min(tid * run, total), w_end = min(w_begin + run, total);
        int cnt = 0;
        for (int i = w_begin; i < w_end; i++) cnt += __popc(sdl_walk_word(occ, H, wpr, i));
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) list_wave_total[wave] = incl;
        __syncthreads();
        int base = incl - cnt;
        for (int w = 0; w < wave; w++) base += list_wave_total[w];
        if (tid == DENSIFY_THREADS - 1) scal[14] = base + cnt;
        for (int i = w_begin; i < w_end; i++) {
            uint32_t bits = sdl_walk_word(occ, H, wpr, i);
            const int y = i / wpr, xb = (i - y * wpr) << 5;
            while (bits) {
                const int x = xb + __ffs((int)bits) - 1;
                bits &= bits - 1u;
                store_wb(sitelist + base++, ((uint32_t)y << 16) | (uint32_t)x);
            }
        }
    }

    // ---- phase C: vertical dilation, in place.  A task = one bitmap column word x 16 rows; every task first
    //      pulls its rows plus the halo into registers, then (after a barrier) stores the ORs.
    {
        const int ntask_rows = (H + MASK_ROWS_PER_TASK - 1) / MASK_ROWS_PER_TASK;
        const int ntasks = ntask_rows * wpr;
        // DENSIFY_THREADS >= ntasks is checked on the host
        uint32_t rows[MASK_ROWS_PER_TASK + 2 * MASK_MAX_HALF];
        const bool active = tid < ntasks;
        const int w = tid % wpr, yb = (tid / wpr) * MASK_ROWS_PER_TASK;
#pragma unroll
        for (int i = 0; i < MASK_ROWS_PER_TASK + 2 * MASK_MAX_HALF; i++) {
            const int y = yb - MASK_MAX_HALF + i;
            rows[i] = (active && y >= 0 && y < H && i >= MASK_MAX_HALF - p && i < MASK_ROWS_PER_TASK + MASK_MAX_HALF + p)
                          ? msk[y * wpr + w]
                          : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MASK_ROWS_PER_TASK; i++) {
            const int y = yb + i;
            uint32_t v = 0;
#pragma unroll
            for (int d = -MASK_MAX_HALF; d <= MASK_MAX_HALF; d++)
                if (d >= -p && d <= p) v |= rows[i + MASK_MAX_HALF + d];
            if (active && y < H) msk[y * wpr + w] = v;
        }
    }
    __syncthreads();

    if (c.out_flags & 2) {
        for (int i = tid; i < H * wpr; i += DENSIFY_THREADS) msk[i] = 0xFFFFFFFFu;
        __syncthreads();
    }
    const int nsites = scal[14];  // sites on the list; scal[0] counts all of them
    // interp_dense_grid_from_sparse early-outs (interpolation_utils.py:39-43): < 4 points, all x equal, all y equal
    const bool degenerate = scal[0] < 4 || scal[1] == scal[2] || scal[3] <= 1;

    // the site list and the base image were written by this workgroup through L2: make them visible to all its waves
    wg_barrier_after_global_stores();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

#if defined(SALVE_PROFILE_WALK)
    SD_PHASE(0, t_phase);
#endif
    // ---- phase E: Delaunay stars.
    //      E1: every lane runs the local state machine (star_local.h) and pulls sites from a shared cursor, so lanes
    //          never idle; owned triangles are queued; sites whose star does not fit the window go to the hard list.
    //      E2: hard sites (hull, sparse regions: a few %) walk their star with the general algorithm and rasterise
    //          in place.   F: all lanes rasterise the queued triangles.
    if (!degenerate && !(dbg_flags & 1)) {
        SdGrid g = {H, W, wpr, occ, rmin, rmax, 0, 1, (dbg_flags & 32) ? nullptr : &d_star_table.off[0][0][0], scal[1], scal[2], scal[12], scal[13], (H <= 1024 && W <= 1024 && !(dbg_flags & 256)) ? tri_cache : nullptr};
        RasterEmit raster = {H, W, wpr, occ, msk, bev, flip, 0, 1, (dbg_flags & 2) != 0, status};
        QueueEmit qemit = {triq, &scal[8], H * W, (dbg_flags & 1024) != 0, status};
        const int hard_cap = (H * W) >> 1;   // entries of two words: the site, and where its lean walk stood
        SdLean st;
        bool active = false, exhausted = false;
        int iters = 0;
        for (;;) {
            // Idle lanes are refilled in batches: the site set-up code is then issued once per ~16 finished sites
            // instead of in nearly every iteration, at the price of a few lane-iterations of idling.
            const unsigned long long idle = __ballot(!active && !exhausted);
            const unsigned long long busy = __ballot(active);
            if (idle != 0ull && (__popcll(idle) >= 16 || busy == 0ull)) {   // (thresholds of 24 ... 48 measured the same or worse, round 3)
                if (!active && !exhausted) {
                    const int i = atomicAdd(&scal[6], 1);
                    if (i >= nsites) {
                        exhausted = true;
                    } else {
                        const uint32_t s = __hip_atomic_load(sitelist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (dbg_stats) { atomicAdd(&scal[9], 1); atomicAdd(&scal[10], (int)(s & 0xFFFFu) + (int)(s >> 16)); }
                        if (sdl_lean_begin(st, g, (int)(s & 0xFFFFu), (int)(s >> 16)) == SDL_LEAN_CONTINUE) active = true;
                        else push_hard(hardlist, &scal[7], hard_cap, s, HARD_FRESH, status);
                    }
                }
            } else if (busy == 0ull) {
                break;  // every lane is exhausted and idle
            }
            if (active) {
                const int r = sdl_lean_step(st, g, qemit);
                iters++;
                if (r != SDL_LEAN_CONTINUE) {
                    active = false;
                    if (dbg_mask && (dbg_flags & 16))  // development: how each site's lean walk ended
                    {
                        dbg_mask[((size_t)rid * H + st.sy) * W + st.sx] = (uint8_t)((r == SDL_LEAN_HARD ? 200 : 100) + min(st.deg, 50));
                    }
                    if (r == SDL_LEAN_HARD)   // the general walk takes over at the edge this walk could not answer
                        push_hard(hardlist, &scal[7], hard_cap, ((uint32_t)st.sy << 16) | (uint32_t)st.sx,
                                  (uint32_t)((st.ax + 128) & 0xFF) | ((uint32_t)((st.ay + 128) & 0xFF) << 8) | ((uint32_t)(st.n0x + 1) << 16) |
                                      ((uint32_t)(st.n0y + 1) << 18) | (st.dir > 0 ? 1u << 20 : 0u) | (st.half ? 1u << 21 : 0u), status);
                }
            }
        }
        if (dbg_stats) atomicAdd(&scal[4], iters);
#if defined(SALVE_PROFILE_WALK)
        SD_PHASE(5, t_phase);   // E1 without the wait for the slowest wave
#endif
        wg_barrier_after_global_stores();  // hard list and triangle queue are complete and in L2
#if defined(SALVE_PROFILE_WALK)
        SD_PHASE(1, t_phase);
#endif
        const int nhard = min(scal[7], hard_cap);
        const int nq = min(scal[8], H * W);
        int err = 0;
#if defined(SALVE_PROFILE_WALK)
        long long t_e2 = SD_NOW();
#endif
        {   // E2: one group of E2_GROUP lanes per hard site; its row sweeps, candidate probes and its rasterisation are shared
            //     by the group's lanes.  The group is the whole wavefront: with several walks side by side in one wavefront
            //     (16- or 32-lane groups) the walks diverge -- one takes a table hit while its neighbour sweeps -- and the
            //     wavefront pays for every path in turn.  Measured, general walk per render (box / cluttered scene):
            //     64 lanes 2.6 / 4.0 us, 32 lanes 2.9 / 4.2 us, 16 lanes 4.0 / 5.1 us, 8 lanes 8.9 / 8.5 us.
            SdGrid gw = g;
            gw.lane = lane & (E2_GROUP - 1);
            gw.nlanes = E2_GROUP;
            gw.gbase = lane & ~(E2_GROUP - 1);
            RasterEmit rw = raster;
            rw.lane = gw.lane;
            rw.nlanes = E2_GROUP;
            // A group takes HARD_RUN consecutive list entries at a time: neighbours in the list are neighbours in the
            // image (the lean walk met them in raster order), and one group walking them one after the other finds the
            // triangles of the previous site in the cache instead of racing another group for them.
            const int nh = (dbg_flags & 4) ? 0 : nhard;
            for (;;) {
                int i0 = 0;
                if (gw.lane == 0) i0 = atomicAdd(&scal[11], HARD_RUN);
                i0 = __shfl(i0, gw.gbase);
                if (i0 >= nh) break;
                // the run's entries in ONE load (lane l holds word l of the run), handed out by lane reads: one trip to L2 per
                // run instead of two dependent ones per site
                static_assert(2 * HARD_RUN <= E2_GROUP, "a run's entries are fetched by one load of the group");
                uint32_t run_words = 0;
                if (gw.lane < 2 * HARD_RUN && 2 * i0 + gw.lane < 2 * nh)
                    run_words = __hip_atomic_load(hardlist + 2 * i0 + gw.lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = i0; i < min(i0 + HARD_RUN, nh); i++) {
                    const uint32_t s = (uint32_t)__shfl((int)run_words, gw.gbase + 2 * (i - i0));
                    const uint32_t w = (uint32_t)__shfl((int)run_words, gw.gbase + 2 * (i - i0) + 1);
                    const int hx = (int)(s & 0xFFFFu), hy = (int)(s >> 16);
                    const int r = sd_walk(gw, hx, hy, (w & HARD_FRESH) != 0, (int)(w & 0xFF) - 128, (int)((w >> 8) & 0xFF) - 128, (w >> 20) & 1u ? 1 : -1,
                                          ((w >> 21) & 1u) != 0, (int)((w >> 16) & 3u) - 1, (int)((w >> 18) & 3u) - 1, rw);
                    if (r < 0) err = 1;
                }
            }
        }
#if defined(SALVE_PROFILE_WALK)
        SD_LAP(e2_total, t_e2);
        SD_PHASE(2, t_phase);
#endif
        if (err) {
            atomicOr(&scal[5], 1);
            // a star walk that did not close: the image of this render is incomplete -- tell the host (salve_hip.h: status word)
            if (status && (lane & (E2_GROUP - 1)) == 0) atomicOr(status, SALVE_STATUS_WALK_FAILED);
        }
        // ---- phase F: the queued triangles.  Two in three have twice-the-area 2: by Pick exactly one lattice point besides the
        //      vertices, the MIDPOINT of their one edge with an even difference vector, colour floor((p + q) / 2); another
        //      fifth has twice-the-area 3 and a lattice centroid, its one interior point, colour floor((a + b + c) / 3).  Those
        //      are filled right here (a dozen instructions each); the general triangles -- a loop over rows and pixels -- are
        //      only listed (their queue index, in the dead site list) and rasterised in a second pass, so that no wavefront
        //      runs at the pace of its general ones.
        uint32_t* genlist = sitelist;
        for (int i = tid; i < ((dbg_flags & 8) ? 0 : nq); i += DENSIFY_THREADS) {
            const unsigned long long e = __hip_atomic_load(triq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ax = (int)(e & 0xFFFFu), ay = (int)((e >> 16) & 0xFFFFu);
            const uint32_t rel = (uint32_t)(e >> 32);
            const int bx = ax + (int8_t)(rel & 0xFF), by = ay + (int8_t)((rel >> 8) & 0xFF), cx = ax + (int8_t)((rel >> 16) & 0xFF), cy = ay + (int8_t)(rel >> 24);
            const int32_t area = sd_orient(ax, ay, bx, by, cx, cy);
            const bool centroid = area == 3 && (ax + bx + cx) % 3 == 0 && (ay + by + cy) % 3 == 0;   // (the other kind of area 3 has two points on one edge: general)
            if (area != 2 && !centroid) {
                store_wb(genlist + atomicAdd(&scal[15], 1), (uint32_t)i);
                continue;
            }
            if (dbg_flags & 2) continue;
            if (area == 2) {
                int px = cx, py = cy, qx = ax, qy = ay;                                  // edge c-a unless ...
                if ((((bx - ax) | (by - ay)) & 1) == 0) { px = ax; py = ay; qx = bx; qy = by; }
                else if ((((cx - bx) | (cy - by)) & 1) == 0) { px = bx; py = by; qx = cx; qy = cy; }
                const int mx = (px + qx) >> 1, my = (py + qy) >> 1;
                if (!(((msk[(my * wpr) + (mx >> 5)] & ~occ[(my * wpr) + (mx >> 5)]) >> (mx & 31)) & 1u)) continue;
                const uint32_t cp = __hip_atomic_load(bev + pixi(px, py), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t cq = __hip_atomic_load(bev + pixi(qx, qy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // floor((p + q) / 2) in each of the three colour bytes at once
                bev[pixi(mx, my)] = (((cp & 0xFEFEFEu) >> 1) + ((cq & 0xFEFEFEu) >> 1) + (cp & cq & 0x010101u));
            } else {
                const int mx = (ax + bx + cx) / 3, my = (ay + by + cy) / 3;
                if (!(((msk[(my * wpr) + (mx >> 5)] & ~occ[(my * wpr) + (mx >> 5)]) >> (mx & 31)) & 1u)) continue;
                const uint32_t ca = __hip_atomic_load(bev + pixi(ax, ay), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t cb = __hip_atomic_load(bev + pixi(bx, by), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t cc = __hip_atomic_load(bev + pixi(cx, cy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t out = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const uint32_t sum = ((ca >> (8 * ch)) & 255u) + ((cb >> (8 * ch)) & 255u) + ((cc >> (8 * ch)) & 255u);
                    out |= ((sum * 0xAAABu) >> 17) << (8 * ch);   // floor(sum / 3), exact below 2^16
                }
                bev[pixi(mx, my)] = out;
            }
        }
        wg_barrier_after_global_stores();   // the list of general triangles is complete and in L2
        const int ngen = min(scal[15], H * W);
        for (int i = tid; i < ngen; i += DENSIFY_THREADS) {
            const uint32_t qi = __hip_atomic_load(genlist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long e = __hip_atomic_load(triq + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ax = (int)(e & 0xFFFFu), ay = (int)((e >> 16) & 0xFFFFu);
            const uint32_t rel = (uint32_t)(e >> 32);
            raster(ax, ay, ax + (int8_t)(rel & 0xFF), ay + (int8_t)((rel >> 8) & 0xFF), ax + (int8_t)((rel >> 16) & 0xFF),
                   ay + (int8_t)(rel >> 24));
        }
    }
    // ---- phase G: data pixels outside the mask are 0 in the result (all of them if the interpolation early-outs).
    //      They kept their colour until here because triangles read their vertex colours from the image.
    wg_barrier_after_global_stores();
#if defined(SALVE_PROFILE_WALK)
    SD_PHASE(3, t_phase);
#endif
    for (int i = tid; i < H * wpr; i += DENSIFY_THREADS) {
        uint32_t bits = occ[i] & (degenerate ? 0xFFFFFFFFu : ~msk[i]);
        const int y = i / wpr, xb = (i % wpr) << 5;
        while (bits) {
            const int x = xb + __ffs((int)bits) - 1;
            bits &= bits - 1u;
            bev[pixi(x, y)] = 0u;
        }
    }
    // ---- phase H (salve_bev_densify_tiles): the render's verifier tile -- Resize (11-bit taps) -> centre crop -> ToTensor -> Normalize of
    //      THIS image and of the pair's pretiled second image, whole 6-channel groups into the fp16 NHWC sample: what
    //      bev_tile_pair_kernel does in a launch of its own, which reads the 1 MB image back from HBM after the whole launch has gone by;
    //      here it comes out of the L2 it was just written to.  The taps and the normalisation table go into the LDS of the (dead) bitmaps.
    if (c.out_flags & DENSIFY_TILES) {
        wg_barrier_after_global_stores();                 // phase G's stores have left the CU (and everyone is done with the bitmaps)
        asm volatile("buffer_inv sc1" ::: "memory");      // the CU's L1 may hold lines of the image as they were BEFORE this workgroup wrote them
        const TileFuse* tfp = reinterpret_cast<const TileFuse*>(reinterpret_cast<const int32_t*>(bitmaps_all + (size_t)gridDim.x * bitmap_words_hw(H, W)) + tile_fuse_word(gridDim.x));
        const TileFuse tf = *tfp;
        const salve_tile_job_t ja = tf.jobs_a[rid], jb = tf.jobs_b[rid];
        if (ja.slot >= 0) {
            int4* cy_l = reinterpret_cast<int4*>(smem);
            int4* cx_l = cy_l + tf.resize;
            float* lut_l = reinterpret_cast<float*>(cx_l + tf.resize);
            for (int i = tid; i < tf.resize; i += DENSIFY_THREADS) {
                cy_l[i] = reinterpret_cast<const int4*>(tf.coef_y)[i];
                cx_l[i] = reinterpret_cast<const int4*>(tf.coef_x)[i];
            }
            for (int i = tid; i < 3 * 256; i += DENSIFY_THREADS) lut_l[i] = tf.lut[i];
            __syncthreads();
            const int crop = tf.crop, off = (tf.resize - crop) / 2, out_c = tf.out_c;
            const bool a_first = ja.chan < jb.chan;
            const int c0 = a_first ? ja.chan : jb.chan;
            const uint32_t* tb = tf.tiles_b + jb.bev_offset;
            uint16_t* osample = tf.out + (size_t)ja.slot * crop * crop * out_c;
            // A thread keeps ONE tile column (its two source columns and their weights stay in registers) and walks the rows: no division per
            // pixel, the row's taps are a broadcast LDS read.  Columns in groups of 256 lanes (crop = 224: 32 idle lanes per group), the
            // DENSIFY_THREADS / 256 groups take the rows in turn.  24-bit multiplies (full rate; the tile kernels' v_mul_lo_u32 are quarter
            // rate): every product here is < 2^28 of operands < 2^16 -- the same integers.
            constexpr int COLS = 256, RGROUPS = DENSIFY_THREADS / COLS;
            for (int j0 = 0; j0 < crop; j0 += COLS) {
                const int j = j0 + (tid & (COLS - 1));
                if (j >= crop) continue;
                const int4 cx = cx_l[j + off];
                // U rows at a time (SALVE_TILE_ROWS_IN_FLIGHT): all their gathers (the image out of the L2, the second image's pixel) are issued before the first is used --
                // one row after the other the phase was a chain of dependent round trips, 112 per thread (+1.0 ms per 4096 renders)
#ifndef SALVE_TILE_ROWS_IN_FLIGHT
#define SALVE_TILE_ROWS_IN_FLIGHT 8   // (r6, same-box A/B: 2 rows 14.62 ms, 4 rows 14.41, 8 rows 14.32 per 4096 renders)
#endif
                constexpr int U = SALVE_TILE_ROWS_IN_FLIGHT;
                for (int i0 = tid / COLS; i0 < crop; i0 += RGROUPS * U) {
                    int4 cy[U];
                    uint32_t p00[U], p01[U], p10[U], p11[U], q[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int i = min(i0 + u * RGROUPS, crop - 1);
                        cy[u] = cy_l[i + off];
                        const uint32_t* r0 = bev + __umul24(cy[u].x, W);
                        const uint32_t* r1 = bev + __umul24(cy[u].y, W);
                        p00[u] = r0[cx.x]; p01[u] = r0[cx.y]; p10[u] = r1[cx.x]; p11[u] = r1[cx.y];
                        q[u] = tb[i * crop + j];
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int i = i0 + u * RGROUPS;
                        if (i >= crop) break;
                        const int idx = i * crop + j;
                        float va[3], vb[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            const uint32_t a = (p00[u] >> (8 * ch)) & 255u, b = (p01[u] >> (8 * ch)) & 255u;
                            const uint32_t cc = (p10[u] >> (8 * ch)) & 255u, d = (p11[u] >> (8 * ch)) & 255u;
                            const int S0 = (int)(__umul24(a, cx.z) + __umul24(b, cx.w));   // horizontal pass, x2048 (tile_pixel, op for op)
                            const int S1 = (int)(__umul24(cc, cx.z) + __umul24(d, cx.w));
                            int r = (int)(((__umul24(cy[u].z, S0 >> 4) >> 16) + (__umul24(cy[u].w, S1 >> 4) >> 16) + 2u) >> 2);
                            r = min(max(r, 0), 255);
                            va[ch] = lut_l[ch * 256 + r];
                            vb[ch] = lut_l[ch * 256 + ((q[u] >> (8 * ch)) & 255u)];
                        }
                        const float* lo = a_first ? va : vb;
                        const float* hi = a_first ? vb : va;
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        const uint32_t w0 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){lo[0], lo[1]}, h2));   // (round to nearest even, as (_Float16)f)
                        const uint32_t w1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){lo[2], hi[0]}, h2));
                        const uint32_t w2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){hi[1], hi[2]}, h2));
                        uint16_t* px = osample + (size_t)idx * out_c;
                        if (out_c == 8) {
                            *reinterpret_cast<uint4*>(px) = make_uint4(w0, w1, w2, 0u);
                        } else {
                            uint32_t* o = reinterpret_cast<uint32_t*>(px + c0);
                            o[0] = w0; o[1] = w1; o[2] = w2;
                            if (c0 + 12 > out_c)
                                for (int k = c0 + 6; k < out_c; k += 2) *reinterpret_cast<uint32_t*>(px + k) = 0u;
                        }
                    }
                }
            }
        }
    }
    if (dbg_mask) {
        for (int i = tid; i < H * W; i += DENSIFY_THREADS) {
            const int y = i / W, x = i % W;
            dbg_mask[(size_t)rid * H * W + i] = (uint8_t)((msk[y * wpr + (x >> 5)] >> (x & 31)) & 1u);
        }
    }
    if (dbg_stats) {
        __syncthreads();
#if defined(SALVE_PROFILE_WALK)
        SD_PHASE(4, t_phase);
        __syncthreads();
        if (tid < 8) dbg_stats[rid * 8 + tid] = (dbg_flags & 128) ? sd_phase[tid] : (dbg_flags & 64) ? sd_timers[tid] : sd_counters[tid];
#else
        if (tid < 8) dbg_stats[rid * 8 + tid] = tid == 1 ? scal[9] : (tid == 2 ? scal[10] : (tid < 6 ? scal[tid] : scal[tid + 1]));  // [1] sites begun [2] checksum [6] hard sites [7] queued triangles
#endif
    }
}

// ------------------------------------------------------------------------------------------------ tiles

__global__ __launch_bounds__(256) void bev_tile_kernel(const uint32_t* __restrict__ bev, int W,
                                                       const salve_tile_job_t* __restrict__ jobs,
                                                       const int32_t* __restrict__ coef_y,
                                                       const int32_t* __restrict__ coef_x, int resize, int crop,
                                                       const float* __restrict__ lut, void* __restrict__ out, int fmt,
                                                       int out_c) {
    const salve_tile_job_t job = jobs[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= crop * crop) return;
    const int i = idx / crop, j = idx % crop;
    const int off = (resize - crop) / 2;
    const int4 cy = reinterpret_cast<const int4*>(coef_y)[i + off];
    const int4 cx = reinterpret_cast<const int4*>(coef_x)[j + off];
    const uint32_t* img = bev + job.bev_offset;
    const uint32_t p00 = img[(size_t)cy.x * W + cx.x], p01 = img[(size_t)cy.x * W + cx.y];
    const uint32_t p10 = img[(size_t)cy.y * W + cx.x], p11 = img[(size_t)cy.y * W + cx.y];
    float v[3];
    int rq[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const int a = (p00 >> (8 * ch)) & 255, b = (p01 >> (8 * ch)) & 255;
        const int cc = (p10 >> (8 * ch)) & 255, d = (p11 >> (8 * ch)) & 255;
        const int S0 = a * cx.z + b * cx.w;  // horizontal pass, x2048
        const int S1 = cc * cx.z + d * cx.w;
        int r = (((cy.z * (S0 >> 4)) >> 16) + ((cy.w * (S1 >> 4)) >> 16) + 2) >> 2;
        r = min(max(r, 0), 255);
        rq[ch] = r;
        v[ch] = lut[ch * 256 + r];
    }
    if (fmt == SALVE_TILE_U8X4) {   // the resized + cropped image itself (before ToTensor / Normalize): 0x00BBGGRR, one image per slot
        reinterpret_cast<uint32_t*>(out)[(size_t)job.slot * crop * crop + idx] = (uint32_t)rq[0] | ((uint32_t)rq[1] << 8) | ((uint32_t)rq[2] << 16);
    } else if (fmt == SALVE_TILE_F32_NCHW) {
        float* o = reinterpret_cast<float*>(out) + ((size_t)job.slot * out_c + job.chan) * crop * crop + idx;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) o[(size_t)ch * crop * crop] = v[ch];
    } else {
        uint16_t* o = reinterpret_cast<uint16_t*>(out) + ((size_t)job.slot * crop * crop + idx) * out_c + job.chan;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) o[ch] = f32_to_f16(v[ch]);
    }
}

// One tile pixel of image `img`: cv2 INTER_LINEAR on uint8 with 11-bit taps, then the normalisation LUT (bev_tile_kernel's arithmetic).
// Both tiles of an early-fusion pair per thread, six channels (plus the sample's zero padding behind its last group) in
// whole-pixel stores: see salve_bev_tile_pairs in salve_hip.h.
__global__ __launch_bounds__(256) void bev_tile_pair_kernel(const uint32_t* __restrict__ bev_a, const uint32_t* __restrict__ bev_b, int W,
                                                            const salve_tile_job_t* __restrict__ jobs_a, const salve_tile_job_t* __restrict__ jobs_b,
                                                            const int32_t* __restrict__ coef_y, const int32_t* __restrict__ coef_x, int resize,
                                                            int crop, const float* __restrict__ lut, uint16_t* __restrict__ out, int out_c,
                                                            int n_pairs, int n_blocks, int xcd_group, int b_pretiled) {
    // (the workgroups of one pair read the same two BEV images: same id % 8 = same XCD = one L2 -- as in bev_splat_kernel)
    int job, blk;
    if (xcd_group) {
        const int id = blockIdx.x, s_ = id >> 3;
        job = (s_ / n_blocks) * 8 + (id & 7);
        blk = s_ % n_blocks;
        if (job >= n_pairs) return;
    } else {
        job = blockIdx.x / n_blocks;
        blk = blockIdx.x % n_blocks;
    }
    const salve_tile_job_t ja = jobs_a[job], jb = jobs_b[job];
    const int idx = blk * 256 + threadIdx.x;
    if (idx >= crop * crop) return;
    const int i = idx / crop, j = idx % crop;
    const int off = (resize - crop) / 2;
    const int4 cy = reinterpret_cast<const int4*>(coef_y)[i + off];
    const int4 cx = reinterpret_cast<const int4*>(coef_x)[j + off];
    float va[3], vb[3];
    tile_pixel(bev_a + ja.bev_offset, W, cy, cx, lut, va);
    if (b_pretiled) {   // image b was resized and cropped before (SALVE_TILE_U8X4, once per panorama): ToTensor + Normalize only
        const uint32_t q = bev_b[jb.bev_offset + idx];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) vb[ch] = lut[ch * 256 + ((q >> (8 * ch)) & 255u)];
    } else {
        tile_pixel(bev_b + jb.bev_offset, W, cy, cx, lut, vb);
    }
    const bool a_first = ja.chan < jb.chan;
    const int c0 = a_first ? ja.chan : jb.chan;   // first of the group's six channels (a multiple of 6)
    uint32_t w[3];                                // the six halves, two per word
    {
        const float* lo = a_first ? va : vb;
        const float* hi = a_first ? vb : va;
        w[0] = (uint32_t)f32_to_f16(lo[0]) | ((uint32_t)f32_to_f16(lo[1]) << 16);
        w[1] = (uint32_t)f32_to_f16(lo[2]) | ((uint32_t)f32_to_f16(hi[0]) << 16);
        w[2] = (uint32_t)f32_to_f16(hi[1]) | ((uint32_t)f32_to_f16(hi[2]) << 16);
    }
    uint16_t* px = out + ((size_t)ja.slot * crop * crop + idx) * out_c;
    if (out_c == 8) {   // one surface: the whole 16-byte pixel, padding channels 6 and 7 included
        *reinterpret_cast<uint4*>(px) = make_uint4(w[0], w[1], w[2], 0u);
    } else {
        uint32_t* o = reinterpret_cast<uint32_t*>(px + c0);   // c0 is even: 4-byte aligned
        o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
        if (c0 + 12 > out_c)   // the sample's last group: zero the padding channels behind it
            for (int c = c0 + 6; c < out_c; c += 2) *reinterpret_cast<uint32_t*>(px + c) = 0u;
    }
}

__global__ __launch_bounds__(256) void bev_export_kernel(const uint32_t* __restrict__ bev, size_t npx,
                                                         uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const uint32_t v = bev[i];
    out[3 * i] = (uint8_t)v;
    out[3 * i + 1] = (uint8_t)(v >> 8);
    out[3 * i + 2] = (uint8_t)(v >> 16);
}

// Panorama ingest: cv2.resize(rgb, (w, h), INTER_LINEAR) on uint8 RGB (bev_rendering_utils.py:370-375).  OpenCV turns an
// exact 2x down-scale into its INTER_AREA fast path ((a + b + c + d + 2) >> 2); everything else is the 11-bit fixed-point
// bilinear of the tile kernel above.  One thread per destination pixel.
__global__ __launch_bounds__(256) void resize_rgb_kernel(const uint8_t* __restrict__ src, int n, int Hs, int Ws,
                                                         uint8_t* __restrict__ dst, int Hd, int Wd,
                                                         const int32_t* __restrict__ coef_y, const int32_t* __restrict__ coef_x,
                                                         int area2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * Hd * Wd) return;
    const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd);
    const uint8_t* img = src + (i / ((size_t)Hd * Wd)) * (size_t)Hs * Ws * 3;
    uint8_t* o = dst + i * 3;
    if (area2) {
        const uint8_t* r0 = img + ((size_t)(2 * y) * Ws + 2 * x) * 3;
        const uint8_t* r1 = r0 + (size_t)Ws * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) o[ch] = (uint8_t)((r0[ch] + r0[3 + ch] + r1[ch] + r1[3 + ch] + 2) >> 2);
        return;
    }
    const int4 cy = reinterpret_cast<const int4*>(coef_y)[y];
    const int4 cx = reinterpret_cast<const int4*>(coef_x)[x];
    const uint8_t* p00 = img + ((size_t)cy.x * Ws + cx.x) * 3;
    const uint8_t* p01 = img + ((size_t)cy.x * Ws + cx.y) * 3;
    const uint8_t* p10 = img + ((size_t)cy.y * Ws + cx.x) * 3;
    const uint8_t* p11 = img + ((size_t)cy.y * Ws + cx.y) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const int S0 = p00[ch] * cx.z + p01[ch] * cx.w;  // horizontal pass, x2048
        const int S1 = p10[ch] * cx.z + p11[ch] * cx.w;
        const int r = (((cy.z * (S0 >> 4)) >> 16) + ((cy.w * (S1 >> 4)) >> 16) + 2) >> 2;
        o[ch] = (uint8_t)min(max(r, 0), 255);
    }
}

// ------------------------------------------------------------------------------------------------ stand-alone utilities
// zorder_utils.choose_elevated_repeated_vals (zorder_utils.py:10-83) for arbitrary slice planes.
__global__ __launch_bounds__(256) void zorder_splat_kernel(const int32_t* __restrict__ x, const int32_t* __restrict__ y,
                                                           const double* __restrict__ z, int n, const double* __restrict__ planes,
                                                           int nslices, int w, int h, unsigned long long* __restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double zi = z[i];
    int slice = -1;
    for (int k = 0; k < nslices; k++)
        if (zi >= planes[k] && zi < planes[k + 1]) slice = k;
    if (slice < 0 || x[i] < 0 || x[i] >= w || y[i] < 0 || y[i] >= h) return;
    atomicMax(img + (size_t)y[i] * w + x[i], ((unsigned long long)(slice + 1) << 32) | (unsigned)i);
}

__global__ __launch_bounds__(256) void zorder_mark_kernel(const int32_t* __restrict__ x, const int32_t* __restrict__ y, int n, int w,
                                                          int h, const unsigned long long* __restrict__ img, uint8_t* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool v = false;
    if (x[i] >= 0 && x[i] < w && y[i] >= 0 && y[i] < h) {
        const unsigned long long k = img[(size_t)y[i] * w + x[i]];
        v = k != 0 && (unsigned)(k & 0xFFFFFFFFull) == (unsigned)i;
    }
    valid[i] = v ? 1 : 0;
}

// interpolation_utils.remove_hallucinated_content (:74-122) on arbitrary H x W x 3 uint8 images and kernel size K.
__global__ __launch_bounds__(256) void halluc_rows_kernel(const uint8_t* __restrict__ sparse, int H, int W, int K,
                                                          uint8_t* __restrict__ rowany) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W), p = K / 2;
    bool any = false;
    for (int xx = x - p; xx < x - p + K && !any; xx++) {
        if (xx < 0 || xx >= W) continue;
        const uint8_t* c = sparse + ((size_t)y * W + xx) * 3;
        any = (((unsigned)c[0] * c[1] * c[2]) & 255u) != 0;  // the uint8 product wraps, as in the reference (:95)
    }
    rowany[i] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void halluc_apply_kernel(const uint8_t* __restrict__ rowany, const uint8_t* __restrict__ interp,
                                                           int H, int W, int K, uint8_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W), p = K / 2;
    bool any = false;
    for (int yy = y - p; yy < y - p + K && !any; yy++)
        if (yy >= 0 && yy < H) any = rowany[(size_t)yy * W + x] != 0;
    for (int ch = 0; ch < 3; ch++) out[3 * i + ch] = any ? interp[3 * i + ch] : 0;
}

// interp_dense_grid_from_sparse (:21-54): sites given as pixel coordinates + colours -> key image (last index wins).
__global__ __launch_bounds__(256) void keys_from_pixels_kernel(const int32_t* __restrict__ xy, int n, int W, int H, uint32_t* __restrict__ kimg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = xy[2 * i], y = xy[2 * i + 1];
    if (x < 0 || x >= W || y < 0 || y >= H) return;
    atomicMax(kimg + (size_t)y * W + x, (1u << KEY_SLICE_SHIFT) | (uint32_t)i);
}

bool make_devcfg(const salve_bev_config_t* cfg, DevCfg* d) {
    if (!cfg) return salve_fail("null config");
    if (cfg->pano_w <= 0 || cfg->pano_w % 4 != 0) return salve_fail("pano_w must be a positive multiple of 4");
    if (cfg->pano_h <= 0 || cfg->crop_rows < 0 || 2 * cfg->crop_rows >= cfg->pano_h) return salve_fail("bad pano_h / crop_rows");
    if (cfg->bev_h < 2 || cfg->bev_w < 2 || cfg->bev_h >= SD_MAX_DIM || cfg->bev_w >= SD_MAX_DIM) return salve_fail("bev size out of range");
    if (cfg->mask_k < 1 || (cfg->mask_k & 1) == 0 || cfg->mask_k / 2 > MASK_MAX_HALF) return salve_fail("mask_k must be odd and <= 17");
    if (cfg->n_slices < 1 || cfg->n_slices > 62) return salve_fail("n_slices out of range");
    if (cfg->out_flags & ~7) return salve_fail("unknown bit in out_flags (1: no flip, 2: no mask, 4: given order): a caller written for another ABI version");
    d->pano_h = cfg->pano_h; d->pano_w = cfg->pano_w; d->crop_rows = cfg->crop_rows;
    d->rows = cfg->pano_h - 2 * cfg->crop_rows;
    d->npts = d->rows * cfg->pano_w;
    if ((long long)d->npts >= (1ll << 21)) return salve_fail("too many pano points for the 21-bit index field");
    d->H = cfg->bev_h; d->W = cfg->bev_w; d->wpr = (cfg->bev_w + 31) / 32; d->mask_half = cfg->mask_k / 2;
    d->depth_scale = cfg->depth_scale;
    d->xmin = cfg->win_xmin; d->xmax = cfg->win_xmax; d->ymin = cfg->win_ymin; d->ymax = cfg->win_ymax;
    d->tx = cfg->img_tx; d->ty = cfg->img_ty; d->scale = cfg->img_scale;
    d->rp00 = cfg->rot_pre[0]; d->rp01 = cfg->rot_pre[1]; d->rp10 = cfg->rot_pre[2]; d->rp11 = cfg->rot_pre[3];
    for (int i = 0; i < 2; i++) { d->zlo[i] = cfg->z_lo[i]; d->zhi[i] = cfg->z_hi[i]; }
    d->zmin = cfg->z_min; d->nslices = cfg->n_slices; d->dbg_flags = cfg->reserved1; d->out_flags = cfg->out_flags;
    const int ntasks = ((d->H + MASK_ROWS_PER_TASK - 1) / MASK_ROWS_PER_TASK) * d->wpr;
    if (ntasks > DENSIFY_THREADS) return salve_fail("bev image too large for the LDS mask pass");
    return true;
}

size_t densify_lds_bytes(const DevCfg& d) {
    return (size_t)2 * d.H * d.wpr * 4 + (size_t)2 * ((d.H + 1) & ~1) * 2 + N_SCAL * 4 + 8 + (size_t)SD_CACHE_SIZE * 8;
}

}  // namespace

// opt in to more than 64 KB of dynamic LDS, for exactly what a launch uses: the attribute is kept per device (a second GPU in
// the process needs its own opt-in) under a mutex
template <class K>
static int ensure_lds(K kernel, size_t lds, size_t* attr_lds /* [64] */, std::mutex& mu) {
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    SALVE_HIP_CHECK(hipGetDevice(&dev));
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    if (dev != slot || lds > attr_lds[slot]) {
        SALVE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds[slot] = lds;
    }
    return SALVE_OK;
}

extern "C" {

// Workspace of n renders: triangle queues (8 B / pixel), site lists, hard-site lists (4 B / pixel each), the splat's bitmap
// words (2 x tiles x 2 KB), and behind them ONE key image for the stand-alone utility paths (an explicit point cloud, explicit
// pixels: single renders).
struct Workspace {
    unsigned long long* triq;
    uint32_t* sitelist;
    uint32_t* hardlist;
    uint32_t* bitmaps;
    uint32_t* keys;   // one image
};

static size_t bitmap_words(const DevCfg& d) { return bitmap_words_hw(d.H, d.W); }

static size_t workspace_per_render(const DevCfg& d) {
    const size_t npx = (size_t)d.H * d.W;
    return npx * (sizeof(unsigned long long) + 2 * sizeof(uint32_t)) + bitmap_words(d) * sizeof(uint32_t);
}

static Workspace carve_workspace(void* workspace, const DevCfg& d, size_t n) {
    const size_t npx = (size_t)d.H * d.W;
    Workspace w;
    w.triq = reinterpret_cast<unsigned long long*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    w.sitelist = reinterpret_cast<uint32_t*>(w.triq + n * npx);
    w.hardlist = w.sitelist + n * npx;
    w.bitmaps = w.hardlist + n * npx;     // (16-byte aligned: npx * 16 bytes per render in front of it, 256-byte base)
    w.keys = w.bitmaps + n * bitmap_words(d);
    return w;
}

size_t salve_bev_workspace_bytes(const salve_bev_config_t* cfg, int32_t n) {
    DevCfg d;
    if (n <= 0 || !make_devcfg(cfg, &d)) return 0;
    return (size_t)n * workspace_per_render(d) + (size_t)d.H * d.W * sizeof(uint32_t) + 256;
}

// ---- panorama index (bev_splat.h): [2 P][entries] float4 boxes, then int32 [2 P] first and [2 P] end of the groups that hold points
size_t salve_bev_pano_index_bytes(const salve_bev_config_t* cfg, int32_t n_panos) {
    DevCfg d;
    if (n_panos <= 0 || !make_devcfg(cfg, &d)) return 0;
    // boxes [2 P][entries] | range_lo [2 P] | range_hi [2 P] | (r6) group boxes [2 P][groups]   (every part a multiple of 16 bytes)
    return (size_t)n_panos * 2 * (pano_grid(d).entries() * sizeof(float4) + 2 * sizeof(int32_t) + (size_t)pano_grid(d).groups() * sizeof(float4));
}

static const int* index_ranges(const DevCfg& d, const void* pano_index, int n_panos) {
    return reinterpret_cast<const int*>(reinterpret_cast<const float4*>(pano_index) + (size_t)n_panos * 2 * pano_grid(d).entries());
}

int salve_bev_pano_index_build(const salve_bev_config_t* cfg, const uint16_t* pano_depth, int32_t n_panos, const double* sphere,
                               void* pano_index, size_t pano_index_bytes, void* stream) {
    DevCfg d;
    if (!make_devcfg(cfg, &d)) return SALVE_ERR_BAD_ARG;
    if (n_panos <= 0 || !pano_depth || !sphere || !pano_index || ((uintptr_t)pano_index & 15)) {
        salve_fail("salve_bev_pano_index_build: null / unaligned pointer or bad count");
        return SALVE_ERR_BAD_ARG;
    }
    if (pano_index_bytes < salve_bev_pano_index_bytes(cfg, n_panos)) { salve_fail("panorama index buffer too small"); return SALVE_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const PanoGrid pg = pano_grid(d);
    int* ranges = const_cast<int*>(index_ranges(d, pano_index, n_panos));
    // ranges start empty: first = 0x7F7F7F7F (atomicMin), end = 0 (atomicMax)
    SALVE_HIP_CHECK(hipMemsetAsync(ranges, 0x7F, (size_t)n_panos * 2 * sizeof(int), s));
    SALVE_HIP_CHECK(hipMemsetAsync(ranges + 2 * n_panos, 0, (size_t)n_panos * 2 * sizeof(int), s));
    const long long waves = (long long)pg.entries() * 2 * n_panos;
    if ((waves + 3) / 4 > 0x7FFFFFFFll) { salve_fail("too many panoramas for one index launch"); return SALVE_ERR_BAD_ARG; }
    hipLaunchKernelGGL(bev_pano_index_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d, pg, pano_depth, sphere,
                       reinterpret_cast<float4*>(pano_index), ranges, ranges + 2 * n_panos, n_panos);
    SALVE_HIP_CHECK(hipGetLastError());
    const long long n_gb = (long long)pg.groups() * 2 * n_panos;   // second level: the union box of every group of 64 blocks
    hipLaunchKernelGGL(bev_pano_group_kernel, dim3((unsigned)((n_gb + 255) / 256)), dim3(256), 0, s, pg, reinterpret_cast<const float4*>(pano_index),
                       reinterpret_cast<float4*>(ranges + 4 * n_panos), 2 * n_panos);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

// does a launch of n renders run its densify stage in cost order?  (both stages ask: the scatter sums the costs, the densify sorts them)
static bool orders_renders(const DevCfg& d, int n, size_t npx) {
    return !(d.out_flags & 4) && n >= ORDER_MIN_RENDERS && (size_t)2 * n * sizeof(int32_t) <= npx * sizeof(uint32_t);
}

static int bev_stage(const salve_bev_config_t* cfg, int stages, const uint8_t* pano_rgb, const uint16_t* pano_depth,
                     int32_t n_panos, const double* sphere, const void* pano_index, const salve_bev_hyp_t* hyps, int32_t n, uint32_t* out_bev,
                     int16_t* dbg_img_xy, uint64_t* dbg_keys, uint8_t* dbg_mask, int32_t* dbg_stats, int32_t* in_window,
                     int32_t* status, void* workspace, size_t workspace_bytes, void* stream, const TileFuse* tiles = nullptr) {
    DevCfg d;
    if (!make_devcfg(cfg, &d)) return SALVE_ERR_BAD_ARG;
    if (n == 0) return SALVE_OK;
    const bool scatter = stages & 1, densify = stages & 2;
    if (n < 0 || !workspace || !out_bev || (scatter && (n_panos <= 0 || !pano_rgb || !pano_depth || !sphere || !pano_index || !hyps))) {
        salve_fail("salve_bev_*: null pointer or bad count");
        return SALVE_ERR_BAD_ARG;
    }
    if (n > 65535) { salve_fail("at most 65535 renders per call"); return SALVE_ERR_BAD_ARG; }
    const size_t need = salve_bev_workspace_bytes(cfg, n);
    if (workspace_bytes < need) { salve_fail("workspace too small"); return SALVE_ERR_WORKSPACE; }
    const size_t lds = densify_lds_bytes(d);
    if (lds > 160 * 1024) { salve_fail("bev image does not fit the 160 KB LDS"); return SALVE_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const size_t npx = (size_t)d.H * d.W;
    const Workspace ws = carve_workspace(workspace, d, (size_t)n);
    const int tiles_x = (d.W + TILE_W - 1) / TILE_W, tiles_y = (d.H + TILE_H - 1) / TILE_H;

    if (scatter) {
        const PanoGrid pg = pano_grid(d);
        const long long n_wg = (long long)((n + 7) / 8 * 8) * tiles_x * tiles_y;
        if (n_wg > 0x7FFFFFFFll) { salve_fail("too many renders for one splat launch"); return SALVE_ERR_BAD_ARG; }
        if (in_window) SALVE_HIP_CHECK(hipMemsetAsync(in_window, 0, (size_t)n * sizeof(int32_t), s));
        // the renders' cost estimates for the densify stage's dispatch order: [cost n][order n] in the key image of the workspace
        int32_t* cost = orders_renders(d, n, npx) ? reinterpret_cast<int32_t*>(ws.keys) : nullptr;
        if (cost) SALVE_HIP_CHECK(hipMemsetAsync(cost, 0, (size_t)n * sizeof(int32_t), s));
        if (dbg_img_xy) SALVE_HIP_CHECK(hipMemsetAsync(dbg_img_xy, 0xFF, (size_t)n * d.npts * 2 * sizeof(int16_t), s));   // (-1, -1): cropped / pruned
        static std::mutex mu;
        static size_t attr[2][64] = {{0}};
        const float4* boxes = reinterpret_cast<const float4*>(pano_index);
        const int* ranges = index_ranges(d, pano_index, n_panos);
        if (dbg_img_xy || dbg_keys) {
            const int st = ensure_lds(bev_splat_kernel<true>, sizeof(SplatLds), attr[1], mu);
            if (st != SALVE_OK) return st;
            hipLaunchKernelGGL((bev_splat_kernel<true>), dim3((unsigned)n_wg), dim3(SPLAT_THREADS), sizeof(SplatLds), s, d, pg, pano_rgb, pano_depth,
                               sphere, hyps, boxes, ranges, ranges + 2 * n_panos, out_bev, ws.bitmaps, in_window, dbg_img_xy, reinterpret_cast<unsigned long long*>(dbg_keys), status, n,
                               n_panos, tiles_x, tiles_y, cost);
        } else {
            const int st = ensure_lds(bev_splat_kernel<false>, sizeof(SplatLds), attr[0], mu);
            if (st != SALVE_OK) return st;
            hipLaunchKernelGGL((bev_splat_kernel<false>), dim3((unsigned)n_wg), dim3(SPLAT_THREADS), sizeof(SplatLds), s, d, pg, pano_rgb, pano_depth,
                               sphere, hyps, boxes, ranges, ranges + 2 * n_panos, out_bev, ws.bitmaps, in_window, nullptr, nullptr, status, n, n_panos, tiles_x, tiles_y, cost);
        }
        SALVE_HIP_CHECK(hipGetLastError());
    }
    if (densify) {
        const int tab_status = ensure_star_table();
        if (tab_status != SALVE_OK) return tab_status;
        static std::mutex mu;
        static size_t attr[2][64] = {{0}};
        DensifyCfg dc = {d.H, d.W, d.wpr, d.mask_half, d.out_flags & 255, d.dbg_flags};
        // the order of the renders inside the launch (bev_order_kernel above); its two int32 arrays live in the workspace's key
        // image, which only the single-render utility paths use
        if (orders_renders(d, n, npx)) {
            int32_t* cost = reinterpret_cast<int32_t*>(ws.keys);   // summed by the scatter stage of the same renders
            hipLaunchKernelGGL(bev_order_kernel, dim3(1), dim3(1024), 0, s, cost, n, cost + n);
            SALVE_HIP_CHECK(hipGetLastError());
            dc.out_flags |= DENSIFY_ORDERED;
        }
        if (tiles) {   // salve_bev_densify_tiles: the tile phase's arguments go into the key image behind order[] (TileFuse)
            if ((tile_fuse_word((size_t)n) * sizeof(int32_t) + sizeof(TileFuse)) > npx * sizeof(uint32_t)) { salve_fail("too many renders for the fused tile phase"); return SALVE_ERR_BAD_ARG; }
            if ((size_t)tiles->resize * 2 * sizeof(int4) + 3 * 256 * sizeof(float) > (size_t)d.H * d.wpr * sizeof(uint32_t)) { salve_fail("resize too large for the fused tile phase"); return SALVE_ERR_UNSUPPORTED; }
            hipLaunchKernelGGL(bev_tile_fuse_kernel, dim3(1), dim3(1), 0, s, *tiles, reinterpret_cast<TileFuse*>(reinterpret_cast<int32_t*>(ws.keys) + tile_fuse_word((size_t)n)));
            SALVE_HIP_CHECK(hipGetLastError());
            dc.out_flags |= DENSIFY_TILES;
        }
        if (dbg_mask || dbg_stats || d.dbg_flags) {   // development outputs or flags: the instantiation that has them
            const int st = ensure_lds(bev_densify_kernel<true>, lds, attr[1], mu);
            if (st != SALVE_OK) return st;
            hipLaunchKernelGGL((bev_densify_kernel<true>), dim3(n), dim3(DENSIFY_THREADS), lds, s, dc, ws.bitmaps, out_bev, ws.sitelist,
                               ws.hardlist, ws.triq, dbg_mask, dbg_stats, (d.dbg_flags & 16) ? dbg_img_xy : nullptr, status);
        } else {
            const int st = ensure_lds(bev_densify_kernel<false>, lds, attr[0], mu);
            if (st != SALVE_OK) return st;
            hipLaunchKernelGGL((bev_densify_kernel<false>), dim3(n), dim3(DENSIFY_THREADS), lds, s, dc, ws.bitmaps, out_bev, ws.sitelist,
                               ws.hardlist, ws.triq, nullptr, nullptr, nullptr, status);
        }
        SALVE_HIP_CHECK(hipGetLastError());
    }
    return SALVE_OK;
}

int salve_bev_render_batch(const salve_bev_config_t* cfg, const uint8_t* pano_rgb, const uint16_t* pano_depth,
                           int32_t n_panos, const double* sphere, const void* pano_index, const salve_bev_hyp_t* hyps, int32_t n,
                           uint32_t* out_bev, int16_t* dbg_img_xy, uint64_t* dbg_keys, uint8_t* dbg_mask,
                           int32_t* dbg_stats, int32_t* out_in_window, int32_t* status, void* workspace, size_t workspace_bytes,
                           void* stream) {
    return bev_stage(cfg, 3, pano_rgb, pano_depth, n_panos, sphere, pano_index, hyps, n, out_bev, dbg_img_xy, dbg_keys, dbg_mask, dbg_stats,
                     out_in_window, status, workspace, workspace_bytes, stream);
}

int salve_bev_scatter(const salve_bev_config_t* cfg, const uint8_t* pano_rgb, const uint16_t* pano_depth, int32_t n_panos,
                      const double* sphere, const void* pano_index, const salve_bev_hyp_t* hyps, int32_t n, uint32_t* out_bev,
                      int16_t* dbg_img_xy, uint64_t* dbg_keys, int32_t* out_in_window, int32_t* status, void* workspace, size_t workspace_bytes,
                      void* stream) {
    return bev_stage(cfg, 1, pano_rgb, pano_depth, n_panos, sphere, pano_index, hyps, n, out_bev, dbg_img_xy, dbg_keys, nullptr, nullptr,
                     out_in_window, status, workspace, workspace_bytes, stream);
}

int salve_bev_densify(const salve_bev_config_t* cfg, int32_t n, uint32_t* out_bev, uint8_t* dbg_mask,
                      int32_t* dbg_stats, int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    return bev_stage(cfg, 2, nullptr, nullptr, 0, nullptr, nullptr, nullptr, n, out_bev, nullptr, nullptr, dbg_mask, dbg_stats,
                     nullptr, status, workspace, workspace_bytes, stream);
}

int salve_bev_densify_tiles(const salve_bev_config_t* cfg, int32_t n, uint32_t* out_bev, const salve_tile_job_t* jobs_a, const salve_tile_job_t* jobs_b,
                            const uint32_t* tiles_b, const int32_t* coef_y, const int32_t* coef_x, int32_t resize, int32_t crop, const float* lut,
                            void* out, int32_t out_c, int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    if (n == 0) return SALVE_OK;
    if (!jobs_a || !jobs_b || !tiles_b || !coef_y || !coef_x || !lut || !out) { salve_fail("salve_bev_densify_tiles: null pointer"); return SALVE_ERR_BAD_ARG; }
    if (crop <= 0 || resize < crop || out_c < 6 || out_c % 2 != 0) { salve_fail("salve_bev_densify_tiles: need 0 < crop <= resize, even out_c >= 6"); return SALVE_ERR_BAD_ARG; }
    const TileFuse t = {jobs_a, jobs_b, tiles_b, coef_y, coef_x, lut, reinterpret_cast<uint16_t*>(out), resize, crop, out_c, 0};
    return bev_stage(cfg, 2, nullptr, nullptr, 0, nullptr, nullptr, nullptr, n, out_bev, nullptr, nullptr, nullptr, nullptr, nullptr, status, workspace,
                     workspace_bytes, stream, &t);
}

// Utility paths: key image 0 -> sparse image + bitmaps of render 0 (the emission of bev_splat_kernel).
static int emit_keys(const DevCfg& d, const Workspace& ws, const uint8_t* rgb, uint32_t* out_bev, uint64_t* dbg_keys, hipStream_t s) {
    static std::mutex mu;
    static size_t attr[64] = {0};
    const int st = ensure_lds(bev_emit_keys_kernel, sizeof(SplatLds), attr, mu);
    if (st != SALVE_OK) return st;
    const int tiles_x = (d.W + TILE_W - 1) / TILE_W, tiles_y = (d.H + TILE_H - 1) / TILE_H;
    hipLaunchKernelGGL(bev_emit_keys_kernel, dim3(tiles_x * tiles_y), dim3(SPLAT_THREADS), sizeof(SplatLds), s, d, ws.keys, rgb, out_bev, ws.bitmaps,
                       reinterpret_cast<unsigned long long*>(dbg_keys), tiles_x, tiles_y);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

int salve_bev_scatter_points(const salve_bev_config_t* cfg, const double* xyz, const uint8_t* rgb, int32_t n_points,
                             uint32_t* out_bev, int32_t* n_in_window, void* workspace, size_t workspace_bytes, void* stream) {
    DevCfg d;
    if (!make_devcfg(cfg, &d)) return SALVE_ERR_BAD_ARG;
    if (n_points < 0 || !workspace || !n_in_window || !out_bev || (n_points > 0 && (!xyz || !rgb))) {
        salve_fail("salve_bev_scatter_points: null pointer or bad count");
        return SALVE_ERR_BAD_ARG;
    }
    if ((long long)n_points >= (1ll << 21)) { salve_fail("at most 2^21 - 1 points"); return SALVE_ERR_UNSUPPORTED; }
    if (workspace_bytes < salve_bev_workspace_bytes(cfg, 1)) { salve_fail("workspace too small"); return SALVE_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const Workspace ws = carve_workspace(workspace, d, 1);
    SALVE_HIP_CHECK(hipMemsetAsync(ws.keys, 0, (size_t)d.H * d.W * sizeof(uint32_t), s));
    SALVE_HIP_CHECK(hipMemsetAsync(n_in_window, 0, sizeof(int32_t), s));
    if (n_points > 0) {
        hipLaunchKernelGGL(bev_scatter_points_kernel, dim3((n_points + 255) / 256), dim3(256), 0, s, d, xyz, n_points, ws.keys, n_in_window);
        SALVE_HIP_CHECK(hipGetLastError());
    }
    return emit_keys(d, ws, rgb, out_bev, nullptr, s);
}

int salve_zorder_winners(const int32_t* x, const int32_t* y, const double* z, int32_t n, const double* planes, int32_t n_slices,
                         int32_t img_w, int32_t img_h, uint64_t* scratch, uint8_t* valid, void* stream) {
    if (n == 0) return SALVE_OK;
    if (n < 0 || !x || !y || !z || !planes || !scratch || !valid || n_slices < 1 || img_w <= 0 || img_h <= 0) {
        salve_fail("salve_zorder_winners: bad argument");
        return SALVE_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    SALVE_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)img_w * img_h * sizeof(uint64_t), s));
    hipLaunchKernelGGL(zorder_splat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, planes, n_slices, img_w, img_h,
                       reinterpret_cast<unsigned long long*>(scratch));
    hipLaunchKernelGGL(zorder_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, img_w, img_h,
                       reinterpret_cast<const unsigned long long*>(scratch), valid);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

int salve_remove_hallucinated(const uint8_t* sparse, const uint8_t* interp, int32_t H, int32_t W, int32_t K, uint8_t* scratch,
                              uint8_t* out, void* stream) {
    if (!sparse || !interp || !scratch || !out || H <= 0 || W <= 0 || K < 1) {
        salve_fail("salve_remove_hallucinated: bad argument");
        return SALVE_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(((long long)H * W + 255) / 256);
    hipLaunchKernelGGL(halluc_rows_kernel, dim3(blocks), dim3(256), 0, s, sparse, H, W, K, scratch);
    hipLaunchKernelGGL(halluc_apply_kernel, dim3(blocks), dim3(256), 0, s, scratch, interp, H, W, K, out);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

int salve_bev_keys_from_pixels(const salve_bev_config_t* cfg, const int32_t* xy, const uint8_t* rgb, int32_t n_points, uint32_t* out_bev,
                               void* workspace, size_t workspace_bytes, void* stream) {
    DevCfg d;
    if (!make_devcfg(cfg, &d)) return SALVE_ERR_BAD_ARG;
    if (n_points < 0 || !workspace || !out_bev || (n_points > 0 && (!xy || !rgb))) { salve_fail("salve_bev_keys_from_pixels: bad argument"); return SALVE_ERR_BAD_ARG; }
    if ((long long)n_points >= (1ll << 21)) { salve_fail("at most 2^21 - 1 points"); return SALVE_ERR_UNSUPPORTED; }
    if (workspace_bytes < salve_bev_workspace_bytes(cfg, 1)) { salve_fail("workspace too small"); return SALVE_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const Workspace ws = carve_workspace(workspace, d, 1);
    SALVE_HIP_CHECK(hipMemsetAsync(ws.keys, 0, (size_t)d.H * d.W * sizeof(uint32_t), s));
    if (n_points > 0) {
        hipLaunchKernelGGL(keys_from_pixels_kernel, dim3((n_points + 255) / 256), dim3(256), 0, s, xy, n_points, d.W, d.H, ws.keys);
        SALVE_HIP_CHECK(hipGetLastError());
    }
    return emit_keys(d, ws, rgb, out_bev, nullptr, s);
}

int salve_bev_export_u8(const uint32_t* bev, int32_t n, int32_t bev_h, int32_t bev_w, uint8_t* out, void* stream) {
    if (n == 0) return SALVE_OK;
    if (!bev || !out || n < 0 || bev_h <= 0 || bev_w <= 0) { salve_fail("salve_bev_export_u8: bad argument"); return SALVE_ERR_BAD_ARG; }
    const size_t npx = (size_t)n * bev_h * bev_w;
    hipLaunchKernelGGL(bev_export_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bev, npx, out);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

int salve_resize_rgb_u8(const uint8_t* src, int32_t n, int32_t src_h, int32_t src_w, uint8_t* dst, int32_t dst_h, int32_t dst_w,
                        const int32_t* coef_y, const int32_t* coef_x, void* stream) {
    if (n == 0) return SALVE_OK;
    if (!src || !dst || n < 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) {
        salve_fail("salve_resize_rgb_u8: null pointer or bad size");
        return SALVE_ERR_BAD_ARG;
    }
    const int area2 = (src_h == 2 * dst_h && src_w == 2 * dst_w) ? 1 : 0;
    if (!area2 && (!coef_y || !coef_x)) { salve_fail("salve_resize_rgb_u8: tap tables are required unless the scale is exactly 2"); return SALVE_ERR_BAD_ARG; }
    const size_t total = (size_t)n * dst_h * dst_w;
    hipLaunchKernelGGL(resize_rgb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, n, src_h, src_w, dst,
                       dst_h, dst_w, coef_y, coef_x, area2);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

int salve_bev_tiles(const uint32_t* bev, int32_t bev_h, int32_t bev_w, const salve_tile_job_t* jobs, int32_t n_jobs,
                    const int32_t* coef_y, const int32_t* coef_x, int32_t resize, int32_t crop, const float* lut,
                    void* out, int32_t out_format, int32_t out_c, void* stream) {
    if (n_jobs == 0) return SALVE_OK;
    if (!bev || !jobs || !coef_y || !coef_x || !lut || !out || n_jobs < 0 || bev_h <= 0 || bev_w <= 0) {
        salve_fail("salve_bev_tiles: null pointer or bad size");
        return SALVE_ERR_BAD_ARG;
    }
    if (crop <= 0 || resize < crop || out_c < 3) { salve_fail("salve_bev_tiles: need 0 < crop <= resize, out_c >= 3"); return SALVE_ERR_BAD_ARG; }
    if (out_format != SALVE_TILE_F32_NCHW && out_format != SALVE_TILE_F16_NHWC && out_format != SALVE_TILE_U8X4) { salve_fail("unknown tile format"); return SALVE_ERR_UNSUPPORTED; }
    if (n_jobs > 65535) { salve_fail("at most 65535 tile jobs per call"); return SALVE_ERR_BAD_ARG; }
    dim3 g((crop * crop + 255) / 256, n_jobs);
    hipLaunchKernelGGL(bev_tile_kernel, g, dim3(256), 0, (hipStream_t)stream, bev, bev_w, jobs, coef_y, coef_x, resize, crop,
                       lut, out, out_format, out_c);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

int salve_bev_tile_pairs(const uint32_t* bev_a, const uint32_t* bev_b, int32_t bev_h, int32_t bev_w, const salve_tile_job_t* jobs_a,
                         const salve_tile_job_t* jobs_b, int32_t n_pairs, const int32_t* coef_y, const int32_t* coef_x, int32_t resize,
                         int32_t crop, const float* lut, void* out, int32_t out_c, int32_t b_pretiled, void* stream) {
    if (n_pairs == 0) return SALVE_OK;
    if (!bev_a || !bev_b || !jobs_a || !jobs_b || !coef_y || !coef_x || !lut || !out || n_pairs < 0 || bev_h <= 0 || bev_w <= 0) {
        salve_fail("salve_bev_tile_pairs: null pointer or bad size");
        return SALVE_ERR_BAD_ARG;
    }
    if (crop <= 0 || resize < crop || out_c < 6 || out_c % 2 != 0) { salve_fail("salve_bev_tile_pairs: need 0 < crop <= resize, even out_c >= 6"); return SALVE_ERR_BAD_ARG; }
    if (n_pairs > 65535) { salve_fail("at most 65535 tile pairs per call"); return SALVE_ERR_BAD_ARG; }
    const int n_blocks = (crop * crop + 255) / 256;
    const int xcd_group = 1;   // the workgroups of a pair on one XCD: its two BEV images are read through one L2 (round 3: -0.3 ms per 4096)
    dim3 g((unsigned)((xcd_group ? (n_pairs + 7) / 8 * 8 : n_pairs) * n_blocks));
    hipLaunchKernelGGL(bev_tile_pair_kernel, g, dim3(256), 0, (hipStream_t)stream, bev_a, bev_b, bev_w, jobs_a, jobs_b, coef_y, coef_x,
                       resize, crop, lut, reinterpret_cast<uint16_t*>(out), out_c, n_pairs, n_blocks, xcd_group, b_pretiled ? 1 : 0);
    SALVE_HIP_CHECK(hipGetLastError());
    return SALVE_OK;
}

}  // extern "C"
