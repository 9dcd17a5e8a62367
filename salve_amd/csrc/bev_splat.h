// bev_splat.h -- front end of the BEV rasteriser (round 4): the z-order key image never exists in HBM.
// Included by bev_render.hip inside its anonymous namespace, after DevCfg and the key constants.
//
//   bev_pano_index_kernel   pose-INDEPENDENT, once per (panorama, surface): the panorama is cut into blocks of 16 x 4 pixels
//                           (one wavefront each); a block's entry is the bounding box, in the frame after the rotmat2d(-90)
//                           product (bev_rendering_utils.py:443-446), of its points that pass the surface's z filter
//                           (:408-413), or an empty box.  Neighbouring panorama pixels see neighbouring surface points, so the
//                           boxes are small (centimetres to decimetres): 90 KB per (panorama, surface) at 1024 x 512.
//   bev_splat_kernel        per (render, 128 x 128 output tile): the tile's z-order keys live in LDS (64 KB).  The workgroup
//                           culls the blocks by their posed boxes (wave ballot -> the compacted list of blocks that can reach
//                           the tile), back-projects only those (the arithmetic of the reference, op for op: :367, :392,
//                           :443-451, :38-45, sim2.py:157-160, np.round), resolves the z-order (zorder_utils.py:49-65) with
//                           ds_max_u32 on (slice + 1) << 21 | raster index, and then emits, with coalesced stores: the tile of
//                           the sparse image (:307-308; the winners' colours gathered from the panorama), and the tile's
//                           occupancy / "non-empty colour" bitmap words, which is all bev_densify_kernel reads.
//                           No key image in HBM, no atomics on memory, no second pass, nothing to zero again.
//   bev_emit_keys_kernel    the same emission for a key image built in memory by the two stand-alone utility paths (an explicit
//                           point cloud / explicit pixels: one render, not on the benchmark's path).
#pragma once

#ifndef SPLAT_THREADS_N
#define SPLAT_THREADS_N 768
#endif
constexpr int SPLAT_THREADS = SPLAT_THREADS_N;
constexpr int TILE_W = 128, TILE_H = 128;   // output tile whose keys live in LDS
constexpr int TILE_LD = TILE_W + 1;         // row stride of the key tile in LDS: with 128 every pixel column would be one bank
constexpr int TILE_WORDS = TILE_W / 32;     // bitmap words per tile row
constexpr int BLK_W = 16, BLK_H = 4;        // panorama block = one wavefront: lane -> (row lane >> 4, column lane & 15)
#ifndef SPLAT_BATCH_N
#define SPLAT_BATCH_N 4
#endif
constexpr int SPLAT_BATCH = SPLAT_BATCH_N;  // blocks a wavefront keeps in flight (loads of all of them issued before any is used)
// (The timing-only switches of this kernel -- cull only / no block loop / no emission / no LDS atomic / no loads -- are a patch,
// tools/probe/ablations/timing_switches.patch, applied by tools/probe/splat_ablation.sh: the product source carries none.)

// Block grid of a panorama: nbr block rows x (gpr groups of 64 block columns); entry (br, g, j) is block column 64 g + j.
struct PanoGrid {
    int nbr, bpr, gpr;
    __host__ __device__ int groups() const { return nbr * gpr; }
    __host__ __device__ size_t entries() const { return (size_t)nbr * gpr * 64; }
};
static inline PanoGrid pano_grid(const DevCfg& d) {
    PanoGrid g;
    g.nbr = (d.rows + BLK_H - 1) / BLK_H;
    g.bpr = (d.pano_w + BLK_W - 1) / BLK_W;
    g.gpr = (g.bpr + 63) / 64;
    return g;
}

// Back-projection of one panorama pixel up to the pre-rotation (bev_scatter_kernel of rounds 1-3, unchanged arithmetic):
// float32 depth product (:367), z = d * zdir(v), x = d * (r(v) cos(theta_u)), y = d * (r(v) sin(theta_u)) (:392,
// hohonet_pano_utils.py:27-43), then xy @ rotmat2d(-90).T in the FMA order of OpenBLAS' dgemm kernel.
struct BackProj {
    double z, x1, y1;
};
__device__ __forceinline__ BackProj back_project(const DevCfg& c, uint32_t dep, double rv, double zv, double ctu, double stu) {
    const float d32 = (float)dep * c.depth_scale;
    const double d = (double)d32;
    BackProj o;
    o.z = d * zv;
    const double x = d * (rv * ctu);
    const double y = d * (rv * stu);
    o.x1 = fma(y, c.rp01, x * c.rp00);
    o.y1 = fma(y, c.rp11, x * c.rp10);
    return o;
}

__global__ __launch_bounds__(256) void bev_pano_index_kernel(DevCfg c, PanoGrid pg, const uint16_t* __restrict__ depth,
                                                             const double* __restrict__ sphere, float4* __restrict__ boxes,
                                                             int* __restrict__ range_lo, int* __restrict__ range_hi, int n_panos) {
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per table entry
    const int lane = threadIdx.x & 63;
    const long long entries = (long long)pg.entries();
    if (wv >= entries * 2 * n_panos) return;
    const int ps = (int)(wv / entries);   // panorama * 2 + surface
    const int e = (int)(wv % entries);
    const int pano = ps >> 1, surface = ps & 1;
    const int br = e / (pg.gpr * 64), bc = e % (pg.gpr * 64);
    const int vr = br * BLK_H + (lane >> 4), u = bc * BLK_W + (lane & 15);
    float xmin = 1e30f, ymin = 1e30f, xmax = -1e30f, ymax = -1e30f;
    if (bc < pg.bpr && vr < c.rows && u < c.pano_w) {
        const int v = vr + c.crop_rows;
        const double* rr = sphere;
        const double* zd = sphere + c.pano_h;
        const double* ct = sphere + 2 * c.pano_h;
        const double* st = ct + c.pano_w;
        const uint32_t dep = depth[((size_t)pano * c.pano_h + v) * c.pano_w + u];
        const BackProj b = back_project(c, dep, rr[v], zd[v], ct[u], st[u]);
        if (b.z > c.zlo[surface] && b.z <= c.zhi[surface]) {
            xmin = xmax = (float)b.x1;
            ymin = ymax = (float)b.y1;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, off)); ymin = fminf(ymin, __shfl_xor(ymin, off));
        xmax = fmaxf(xmax, __shfl_xor(xmax, off)); ymax = fmaxf(ymax, __shfl_xor(ymax, off));
    }
    if (lane == 0) {
        boxes[wv] = make_float4(xmin, ymin, xmax, ymax);
        if (xmin <= xmax) {   // the range of groups that hold any point: [range_lo, range_hi)
            atomicMin(range_lo + ps, e >> 6);
            atomicMax(range_hi + ps, (e >> 6) + 1);
        }
    }
}

// (r6) Second level of the index: the union box of every GROUP of 64 blocks.  A tile's workgroup tests the group boxes first -- all of them in one
// pass, a box per thread -- and its wavefronts then visit only the groups that can reach the tile; before, every wavefront walked the whole
// range of groups that hold points, one dependent LDS atomic + 1 KB box load + ballot per group (2048 x 1024 panoramas: 176 visits per tile and
// render, 20 % of the kernel's time; a tile is reached by a sixth of them).
__global__ __launch_bounds__(256) void bev_pano_group_kernel(PanoGrid pg, const float4* __restrict__ boxes, float4* __restrict__ gboxes, int n_ps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long groups = pg.groups();
    if (i >= groups * n_ps) return;
    const float4* b = boxes + (size_t)(i / groups) * pg.entries() + (size_t)(i % groups) * 64;
    float xmin = 1e30f, ymin = 1e30f, xmax = -1e30f, ymax = -1e30f;
    for (int j = 0; j < 64; j++) {
        const float4 v = b[j];
        if (v.x <= v.z) { xmin = fminf(xmin, v.x); ymin = fminf(ymin, v.y); xmax = fmaxf(xmax, v.z); ymax = fmaxf(ymax, v.w); }
    }
    gboxes[i] = make_float4(xmin, ymin, xmax, ymax);
}

// can the box (pre-rotated frame) reach the tile [wx0, wx1] x [wy0, wy1] (posed frame) under the pose?  float32, conservative: centre + |R| half-extents
// + the rounding slack of the test itself; an empty box (x > z) reaches nothing
__device__ __forceinline__ bool splat_reaches(const float4 bb, float fR00, float fR01, float fR10, float fR11, float ftx, float fty, float wx0,
                                              float wx1, float wy0, float wy1) {
    const float cx = 0.5f * (bb.x + bb.z), cy = 0.5f * (bb.y + bb.w), hx = 0.5f * (bb.z - bb.x), hy = 0.5f * (bb.w - bb.y);
    const float px = fR00 * cx + fR01 * cy + ftx, py = fR10 * cx + fR11 * cy + fty;
    const float qx = fabsf(fR00) * hx + fabsf(fR01) * hy, qy = fabsf(fR10) * hx + fabsf(fR11) * hy;
    const float slack = 1e-5f * (fabsf(cx) + fabsf(cy) + hx + hy + fabsf(ftx) + fabsf(fty));
    return bb.x <= bb.z && px + qx + slack >= wx0 && px - qx - slack <= wx1 && py + qy + slack >= wy0 && py - qy - slack <= wy1;
}

// LDS of the splat / emit kernels: key tile, the tile's bitmap words, two counters.
constexpr int GLIST_CAP = 1024;   // groups of 64 blocks a tile's pre-cull can list (2048 x 1024 panoramas: 352 groups per surface)
struct SplatLds {
    uint32_t tile[TILE_H * TILE_LD];
    uint32_t bm[2][TILE_H][TILE_WORDS];
    int next_group, in_window;
    int n_list;                 // (r6) groups whose union box can reach this tile ...
    int glist[GLIST_CAP];       // ... listed by the pre-cull, in any order (winners are maxima: order-independent)
};

// Emission of a finished key tile: sparse-image tile + bitmap words (layout of the bitmaps of one render:
// [occupancy | non-empty][tile][TILE_H rows][TILE_WORDS], so that a tile's words are one contiguous 2 KB piece).
template <bool DEV>
__device__ __forceinline__ void emit_tile(const DevCfg& c, SplatLds& s, const uint8_t* __restrict__ colours, uint32_t* __restrict__ bev,
                                          uint32_t* __restrict__ bitmaps, unsigned long long* __restrict__ dbg_keys, int t, int ntiles,
                                          int tx0, int ty0, int32_t* __restrict__ cost = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = SPLAT_THREADS >> 6;
    const int flip = (c.out_flags & 1) ? -1 : c.H - 1;
    // EMIT_ROWS rows x two 64-pixel segments per step: all keys, then all colour gathers (a fifth of the pixels hold a winner),
    // then the stores -- the step is one trip to memory, not one per row
    constexpr int EMIT_ROWS = 4, NQ = 2 * EMIT_ROWS;
    for (int r0 = wave * EMIT_ROWS; r0 < TILE_H; r0 += nwaves * EMIT_ROWS) {
        uint32_t key[NQ], col[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) key[q] = s.tile[(r0 + (q >> 1)) * TILE_LD + (q & 1) * 64 + lane];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            col[q] = 0u;
            if (key[q] != 0u) {   // the winner's colour from its source array (:307-308: rgb * 255 truncated == the source uint8)
                const uint8_t* cs = colours + 3 * (size_t)(key[q] & KEY_INDEX_MASK);
                col[q] = (uint32_t)cs[0] | ((uint32_t)cs[1] << 8) | ((uint32_t)cs[2] << 16);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int row = r0 + (q >> 1), x = tx0 + (q & 1) * 64 + lane, y = ty0 + row;
            if (x < c.W && y < c.H) {
                const uint32_t p = (uint32_t)((flip >= 0 ? flip - y : y) * c.W + x);
                bev[p] = col[q];
                if (DEV && dbg_keys)
                    dbg_keys[(size_t)y * c.W + x] = key[q] ? (((unsigned long long)(key[q] >> KEY_SLICE_SHIFT) << 45) |
                                                              ((unsigned long long)(key[q] & KEY_INDEX_MASK) << 24) | col[q]) : 0ull;
            }
            const bool site = key[q] != 0u;
            const uint32_t r = col[q] & 255u, g = (col[q] >> 8) & 255u, b = (col[q] >> 16) & 255u;
            const bool ne = site && (((r * g * b) & 255u) != 0u);   // the uint8 channel product wraps (interpolation_utils.py:95)
            const unsigned long long ob = __ballot(site), nb = __ballot(ne);
            if (lane == 0) {
                s.bm[0][row][(q & 1) * 2] = (uint32_t)ob; s.bm[0][row][(q & 1) * 2 + 1] = (uint32_t)(ob >> 32);
                s.bm[1][row][(q & 1) * 2] = (uint32_t)nb; s.bm[1][row][(q & 1) * 2 + 1] = (uint32_t)(nb >> 32);
            }
        }
    }
    __syncthreads();
    static_assert(TILE_WORDS == 4 && 2 * TILE_H <= SPLAT_THREADS, "one 16-byte store per tile row and bitmap");
    if (tid < 2 * TILE_H) {
        const int which = tid / TILE_H, row = tid % TILE_H;
        const uint4 v = *reinterpret_cast<const uint4*>(&s.bm[which][row][0]);
        *reinterpret_cast<uint4*>(bitmaps + (((size_t)which * ntiles + t) * TILE_H + row) * TILE_WORDS) = v;
        // The render's cost estimate for the densify stage's dispatch order (bev_render.hip: bev_order_kernel -- the costly renders
        // of a launch first): the number of sites with two or more of their four neighbours missing, counted tile by tile from the
        // occupancy words that are in LDS right here (rows and columns beyond the tile count as present).
        static_assert(TILE_H % 64 == 0, "whole wavefronts count the occupancy rows");
        if (cost && which == 0) {   // (wave-uniform: TILE_H is a multiple of 64)
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t up = row > 0 ? s.bm[0][row - 1][k] : ~0u, dn = row < TILE_H - 1 ? s.bm[0][row + 1][k] : ~0u;
                const uint32_t R = (w[k] >> 1) | (k < 3 ? w[k + 1] << 31 : 0x80000000u), L = (w[k] << 1) | (k > 0 ? w[k - 1] >> 31 : 1u);
                const uint32_t three = (L & R & (up | dn)) | (up & dn & (L | R));   // at least three of the four neighbours are sites
                cnt += __popc(w[k] & ~three);
            }
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
            if (lane == 0 && cnt) atomicAdd(cost, cnt);
        }
    }
}

template <bool DEV>
__global__ __launch_bounds__(SPLAT_THREADS, SPLAT_THREADS / 128) void bev_splat_kernel(
    DevCfg c, PanoGrid pg, const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth, const double* __restrict__ sphere,
    const salve_bev_hyp_t* __restrict__ hyps, const float4* __restrict__ boxes, const int* __restrict__ range_lo, const int* __restrict__ range_hi,
    uint32_t* __restrict__ bev_all, uint32_t* __restrict__ bitmaps_all, int32_t* __restrict__ in_window, int16_t* __restrict__ dbg_xy_arg,
    unsigned long long* __restrict__ dbg_keys_arg, int32_t* __restrict__ status, int n_renders, int n_panos, int tiles_x, int tiles_y,
    int32_t* __restrict__ cost) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SplatLds& s = *reinterpret_cast<SplatLds*>(smem);
    int16_t* const dbg_xy = DEV ? dbg_xy_arg : nullptr;
    unsigned long long* const dbg_keys = DEV ? dbg_keys_arg : nullptr;
    // Workgroup -> (render, tile): consecutive workgroup ids are dealt round-robin to the 8 XCDs; all tiles of a render carry
    // the same id % 8, so a render's panorama (depth blocks, box table, colours) is read through ONE L2 (speed only).
    const int ntiles = tiles_x * tiles_y;
    const int id = blockIdx.x, s_ = id >> 3;
    const int rid = (s_ / ntiles) * 8 + (id & 7);
    const int t = s_ % ntiles;
    if (rid >= n_renders) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const salve_bev_hyp_t h = hyps[rid];
    const int tx0 = (t % tiles_x) * TILE_W, ty0 = (t / tiles_x) * TILE_H;
    const int tw = min(TILE_W, c.W - tx0), th = min(TILE_H, c.H - ty0);

    static_assert((TILE_H * TILE_LD) % 4 == 0, "the key tile is zeroed in 16-byte pieces, the bitmap words behind it are read in such");
    for (int i = tid; i < TILE_H * TILE_LD / 4; i += SPLAT_THREADS) reinterpret_cast<uint4*>(s.tile)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) { s.next_group = 0; s.in_window = 0; s.n_list = 0; }

    // The tile in the posed frame: pixel index rint((x + tx) * scale) in [tx0, tx0 + tw) <=> x in [(tx0 - .5) / scale - tx,
    // (tx0 + tw - .5) / scale - tx]; border tiles reach to the window's edge (they own the points whose index clamps to them),
    // nothing reaches beyond the window (prune_to_2d_bbox, :38-45).  float32 with a margin: the test only has to be conservative.
    const float margin = 0.02f;
    float wx0 = (float)fmax(tx0 == 0 ? c.xmin : ((double)tx0 - 0.5) / c.scale - c.tx, c.xmin) - margin;
    float wx1 = (float)fmin(tx0 + tw == c.W ? c.xmax : ((double)(tx0 + tw) - 0.5) / c.scale - c.tx, c.xmax) + margin;
    float wy0 = (float)fmax(ty0 == 0 ? c.ymin : ((double)ty0 - 0.5) / c.scale - c.ty, c.ymin) - margin;
    float wy1 = (float)fmin(ty0 + th == c.H ? c.ymax : ((double)(ty0 + th) - 0.5) / c.scale - c.ty, c.ymax) + margin;
    const float fR00 = h.apply_pose ? h.R[0] : 1.f, fR01 = h.apply_pose ? h.R[1] : 0.f, fR10 = h.apply_pose ? h.R[2] : 0.f,
                fR11 = h.apply_pose ? h.R[3] : 1.f;
    const float ftx = h.apply_pose ? h.t[0] * 1.5f : 0.f, fty = h.apply_pose ? h.t[1] * 1.5f : 0.f;

    // a row that names a panorama outside the batch or an unknown surface renders as an empty image, and says so
    const bool bad_row = h.pano_idx < 0 || h.pano_idx >= n_panos || (unsigned)h.surface > 1u;
    if (bad_row && status && t == 0 && tid == 0) atomicOr(status, SALVE_STATUS_BAD_HYPOTHESIS);
    const int ps = bad_row ? 0 : h.pano_idx * 2 + h.surface;
    const float4* bx = boxes + (size_t)ps * pg.entries();
    const int g_lo = bad_row ? 0 : range_lo[ps], g_hi = bad_row ? 0 : range_hi[ps];

    const double* rr = sphere;
    const double* zd = sphere + c.pano_h;
    const double* ct = sphere + 2 * c.pano_h;
    const double* st = ct + c.pano_w;
    const double zlo = c.zlo[h.surface], zhi = c.zhi[h.surface];
    const double R00 = (double)h.R[0], R01 = (double)h.R[1], R10 = (double)h.R[2], R11 = (double)h.R[3];
    const double ptx = (double)(h.t[0] * 1.5f), pty = (double)(h.t[1] * 1.5f);   // float32 product, then widened (:451)
    const uint16_t* dpano = depth + (bad_row ? (size_t)0 : (size_t)h.pano_idx * c.pano_h * c.pano_w);
    int my_in_window = 0;
#define SPLAT_REACHES(BB) splat_reaches((BB), fR00, fR01, fR10, fR11, ftx, fty, wx0, wx1, wy0, wy1)
    __syncthreads();
    // ---- pre-cull (r6): the groups whose UNION box reaches the tile, a group per thread, listed in LDS (bev_pano_group_kernel)
    const bool listed = g_hi - g_lo <= GLIST_CAP;
    if (listed) {
        const float4* gb = reinterpret_cast<const float4*>(range_lo + 4 * n_panos) + (size_t)ps * pg.groups();
        for (int i = g_lo + tid; i < g_hi; i += SPLAT_THREADS)
        {
            const float4 gbb = gb[i];
            if (SPLAT_REACHES(gbb)) s.glist[atomicAdd(&s.n_list, 1)] = i;
        }
        __syncthreads();
    }
    const int n_visit = listed ? s.n_list : g_hi - g_lo;

    for (;;) {
        // a wavefront takes one group of 64 blocks at a time (dynamic: the groups that reach a tile are few and uneven)
        int g = 0;
        if (lane == 0) {
            g = atomicAdd(&s.next_group, 1);
            g = g < n_visit ? (listed ? s.glist[g] : g_lo + g) : -1;
        }
        g = __shfl(g, 0);
        if (g < 0) break;
        const float4 bb = bx[(size_t)g * 64 + lane];
        const bool hit = SPLAT_REACHES(bb);
        unsigned long long m = __ballot(hit);
        if (m == 0ull) continue;
        const int br = g / pg.gpr, gc = g - br * pg.gpr;
        const int vr = br * BLK_H + (lane >> 4);                 // row of this lane's pixel in every block of the group
        const bool row_ok = vr < c.rows;
        const int v = min(vr, c.rows - 1) + c.crop_rows;
        const double rv = rr[v], zv = zd[v];
        while (m != 0ull) {
            int u[SPLAT_BATCH];
            uint32_t dep[SPLAT_BATCH];
            double ctu[SPLAT_BATCH], stu[SPLAT_BATCH];
            bool ok[SPLAT_BATCH];
#pragma unroll
            for (int k = 0; k < SPLAT_BATCH; k++) {   // (wave-uniform control: m is a ballot)
                ok[k] = false;
                u[k] = 0;
                if (m != 0ull) {
                    const int bit = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    u[k] = (gc * 64 + bit) * BLK_W + (lane & 15);
                    ok[k] = row_ok && u[k] < c.pano_w;
                }
            }
            // (32-bit element offsets from wave-uniform bases: one address register per load instead of a 64-bit sum)
            const uint32_t vrow = (uint32_t)v * (uint32_t)c.pano_w;
#pragma unroll
            for (int k = 0; k < SPLAT_BATCH; k++) {
                const uint32_t uu = ok[k] ? (uint32_t)u[k] : 0u;   // lanes without a pixel read pixel (v, 0): valid memory, result unused
                dep[k] = dpano[vrow + uu]; ctu[k] = ct[uu]; stu[k] = st[uu];
            }
#pragma unroll
            for (int k = 0; k < SPLAT_BATCH; k++) {
                // branch-free up to the stores: a wavefront executes every path any of its lanes takes anyway
                const BackProj b = back_project(c, dep[k], rv, zv, ctu[k], stu[k]);
                bool live = ok[k] && b.z > zlo && b.z <= zhi;
                double x1 = b.x1, y1 = b.y1;
                if (h.apply_pose) {   // (wave-uniform) xy @ R32.T + t32 * 1.5 (:448-451), OpenBLAS' FMA order
                    const double x2 = fma(y1, R01, x1 * R00) + ptx;
                    const double y2 = fma(y1, R11, x1 * R10) + pty;
                    x1 = x2; y1 = y2;
                }
                live = live && c.xmin <= x1 && x1 <= c.xmax && c.ymin <= y1 && y1 <= c.ymax;
                // bevimg_Sim2_world.transform_from: (p @ I.T + t) * s, then np.round (half to even); the identity product is exact
                // up to the sign of a zero, which the rounding erases
                const int ix = live ? (int)rint((x1 + c.tx) * c.scale) : -1;
                const int iy = live ? (int)rint((y1 + c.ty) * c.scale) : -1;
                // the tile that owns the point (counts it, reports it): the one its clamped index falls into
                const int ox = min(max(ix, 0), c.W - 1) - tx0, oy = min(max(iy, 0), c.H - 1) - ty0;
                live = live && (unsigned)ox < (unsigned)TILE_W && (unsigned)oy < (unsigned)TILE_H;
                my_in_window += live ? 1 : 0;
                const uint32_t p = (uint32_t)vr * (uint32_t)c.pano_w + (uint32_t)u[k];   // raster index in the cropped panorama
                if (DEV && dbg_xy && live) {
                    int16_t* o = dbg_xy + ((size_t)rid * c.npts + p) * 2;
                    o[0] = (int16_t)ix; o[1] = (int16_t)iy;
                }
                const double zs = floor(b.z) - c.zmin;   // unit slices from an integer z_min: exact (zorder_utils.py:49-59)
                const bool splat = live && zs >= 0.0 && zs < (double)c.nslices && ix >= 0 && ix < c.W && iy >= 0 && iy < c.H;
                if (splat)
                    atomicMax(&s.tile[(iy - ty0) * TILE_LD + (ix - tx0)], ((uint32_t)((int)zs + 1) << KEY_SLICE_SHIFT) | p);
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) my_in_window += __shfl_xor(my_in_window, off);
    if (lane == 0 && my_in_window) atomicAdd(&s.in_window, my_in_window);
    __syncthreads();
    if (tid == 0 && in_window && s.in_window) atomicAdd(in_window + rid, s.in_window);
    const uint8_t* colours = rgb + ((size_t)h.pano_idx * c.pano_h + c.crop_rows) * c.pano_w * 3;
    emit_tile<DEV>(c, s, colours, bev_all + (size_t)rid * c.H * c.W, bitmaps_all + (size_t)rid * 2 * ntiles * TILE_H * TILE_WORDS,
                   dbg_keys ? dbg_keys + (size_t)rid * c.H * c.W : nullptr, t, ntiles, tx0, ty0, cost ? cost + rid : nullptr);
}

// Utility paths (one render): key image in memory -> the same emission.
__global__ __launch_bounds__(SPLAT_THREADS) void bev_emit_keys_kernel(DevCfg c, const uint32_t* __restrict__ keys, const uint8_t* __restrict__ colours,
                                                                      uint32_t* __restrict__ bev, uint32_t* __restrict__ bitmaps,
                                                                      unsigned long long* __restrict__ dbg_keys, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SplatLds& s = *reinterpret_cast<SplatLds*>(smem);
    const int t = blockIdx.x, tx0 = (t % tiles_x) * TILE_W, ty0 = (t / tiles_x) * TILE_H;
    for (int i = threadIdx.x; i < TILE_H * TILE_W; i += SPLAT_THREADS) {
        const int lx = i % TILE_W, ly = i / TILE_W, x = tx0 + lx, y = ty0 + ly;
        s.tile[ly * TILE_LD + lx] = (x < c.W && y < c.H) ? keys[(size_t)y * c.W + x] : 0u;
    }
    __syncthreads();
    emit_tile<true>(c, s, colours, bev, bitmaps, dbg_keys, t, tiles_x * tiles_y, tx0, ty0);
}
