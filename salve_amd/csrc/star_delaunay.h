// star_delaunay.h -- Delaunay stars of lattice sites, one site at a time, straight from an occupancy bitmap.
//
// Replaces the scipy/Qhull triangulation behind the reference's densification step
//   salve/utils/interpolation_utils.py:46-48  (scipy.interpolate.griddata, method="linear").
//
// Why this shape on MI355X: a global triangulation needs a mutable mesh (2 n triangles x adjacency, ~1.6 MB
// per render) plus rounds of conflict-resolved flips.  Here the only state is the 501x501 occupancy BITMAP
// (32 KB, lives in LDS) and every site computes its own Delaunay star independently by gift-wrapping:
// no atomics, no mesh, no grid-wide rounds, and the integer predicates are exact (all sites are lattice points,
// |coordinate| < 2048, so orient2d fits int32 and the in-circle determinant fits int64).
//
// Uniqueness: co-circular sites are everywhere on a lattice.  The triangulation is made unique by the symbolic
// perturbation z_i = x_i^2 + y_i^2 + eps_i with eps_i >> eps_j > 0 whenever site i precedes site j in raster
// order (y, then x) -- the same rule the CPU oracle uses, so any two correct algorithms agree triangle for triangle.
//
// The header is plain C++ (no HIP types) so that the host-side unit tests can compile the very same code with g++.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define SD_FN __host__ __device__ __forceinline__
#else
#define SD_FN static inline
#endif

#define SD_MAX_DIM 2048  // coordinates must be < 2048 for the int32 / int64 predicate bounds below

// Products of coordinates, coordinate differences (|.| < 2^12) and bitmap row indices fit 24-bit operands: v_mul_i32_i24 runs at
// the full vector rate, the 32-bit v_mul_lo_u32 the compiler must assume at a quarter of it -- and the predicates, the lambda of
// a sweep candidate and every bitmap probe are made of these (round 3: 167 of them in the densify kernel).
// (SD_MUL_DEVICE: __mul24 or a plain product -- the 24-bit form moved the kernel's SGPR spills INTO phase B's row loop in
// three of four placements tried, which cost more than the multiplies saved)
#ifndef SD_MUL_DEVICE
#define SD_MUL_DEVICE(a, b) ((a) * (b))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SD_MUL(a, b) SD_MUL_DEVICE((a), (b))
#else
#define SD_MUL(a, b) ((a) * (b))
#endif

struct SdGrid {
    int H, W, wpr;        // image height, width, 32-bit words per bitmap row
    const uint32_t* occ;  // [H][wpr] occupancy bits, bit (x & 31) of word x >> 5
    const int16_t* rmin;  // [H] smallest occupied x of the row, or W if the row is empty
    const int16_t* rmax;  // [H] largest occupied x of the row, or -1
    // Row sweeps can be shared by the lanes of a wavefront that all work on the SAME site: this caller then visits
    // rows lane, lane + nlanes, ... of every sweep and the lanes merge their candidates (sd_share_best).
    int lane, nlanes;     // 0, 1 for a caller that sweeps alone
    const int8_t* tab;    // SdTable (star_table.h) or nullptr: apex candidates of short edges, best first
    int bx0, bx1, by0, by1;  // bounding box of all sites (inclusive)
    // Optional cache of resolved Delaunay triangles, shared by all callers working on this grid (LDS in the kernel):
    // directed edge (u -> v) -> the apex on its left, one 64-bit word per entry, direct-mapped, SD_CACHE_SIZE entries
    // (0 = empty).  Every triangle found serves three directed edges; along the outline of the point cloud, where the
    // general walk is needed, neighbouring sites ask for the same triangles: about half of the sweeps are avoided.
    unsigned long long* cache;  // or nullptr; requires H, W <= 1024
    // The lanes that share a walk are an aligned group of nlanes (a power of two) consecutive lanes of the wavefront
    // starting at lane gbase -- the whole wavefront (64, 0) or a part of it, several walks per wavefront then.
    int gbase;
};

#define SD_CACHE_SIZE 1536

SD_FN unsigned long long sd_cache_key(int ux, int uy, int vx, int vy) {
    return (unsigned long long)ux | ((unsigned long long)uy << 10) | ((unsigned long long)vx << 20) | ((unsigned long long)vy << 30);
}
SD_FN int sd_cache_slot(unsigned long long key) {
    const unsigned long long h = (key * 0x9E3779B97F4A7C15ull) >> 32;  // 32 well-mixed bits -> [0, SD_CACHE_SIZE)
    return (int)((h * (unsigned long long)SD_CACHE_SIZE) >> 32);
}

// apex on the left of u -> v, if some walk has already resolved that triangle: 1 = found, 2 = u -> v is known to be a hull edge
// (nothing on its left: bit 63 of the entry; every hull edge is asked for twice, once from either end), 0 = not in the cache
#define SD_CACHE_HULL (1ull << 63)
SD_FN int sd_cache_lookup(const SdGrid& g, int ux, int uy, int vx, int vy, int* px, int* py) {
    if (g.cache == nullptr) return 0;
    const unsigned long long key = sd_cache_key(ux, uy, vx, vy);
    const unsigned long long e = g.cache[sd_cache_slot(key)];
    if (((e & ~SD_CACHE_HULL) >> 20) != key || e == 0ull) return 0;
    if (e & SD_CACHE_HULL) return 2;
    *px = (int)(e & 1023u);
    *py = (int)((e >> 10) & 1023u);
    return 1;
}
SD_FN void sd_cache_insert_hull(const SdGrid& g, int ux, int uy, int vx, int vy) {
    if (g.cache == nullptr || g.lane != 0) return;
    const unsigned long long k0 = sd_cache_key(ux, uy, vx, vy);
    g.cache[sd_cache_slot(k0)] = (k0 << 20) | SD_CACHE_HULL;
}

// triangle (u, v, p), counter-clockwise: three directed edges, each with the third vertex on its left.  One lane
// writes; an 8-byte LDS store is indivisible, so a concurrent reader sees an old or a new entry, never a mixture.
SD_FN void sd_cache_insert(const SdGrid& g, int ux, int uy, int vx, int vy, int px, int py) {
    if (g.cache == nullptr || g.lane != 0) return;
    const unsigned long long k0 = sd_cache_key(ux, uy, vx, vy), k1 = sd_cache_key(vx, vy, px, py), k2 = sd_cache_key(px, py, ux, uy);
    g.cache[sd_cache_slot(k0)] = (k0 << 20) | (unsigned long long)px | ((unsigned long long)py << 10);
    g.cache[sd_cache_slot(k1)] = (k1 << 20) | (unsigned long long)ux | ((unsigned long long)uy << 10);
    g.cache[sd_cache_slot(k2)] = (k2 << 20) | (unsigned long long)vx | ((unsigned long long)vy << 10);
}

// Table of apex candidates for short edges (built by star_table.h): for the edge vector (ax, ay) from the origin, the
// lattice points strictly left of the edge in the order in which a circle through the edge's end points, grown to the
// left, meets them (ties by the perturbation).  The first occupied one is the apex.
#define SDT_AMAX 5                                   // edge vectors with |ax|, |ay| <= SDT_AMAX
#define SDT_SIDE (2 * SDT_AMAX + 1)
#define SDT_NVEC (SDT_SIDE * SDT_SIDE)
// Candidates kept per vector.  The LEAN walk (one site per lane, four candidates per step) uses the first SDT_LEAN_LEN of them
// (measured: 32 -> 64 removes 13 % of the hard sites, 128 adds nothing); the GENERAL walk probes 64 per round with its whole
// wavefront, so a long table is cheap there -- and an apex found in the table needs no sweep at all: the table order IS the exact
// circle-growth order (round 4).
#ifndef SDT_LEN
#define SDT_LEN 512
#endif
#define SDT_LEAN_LEN 64
#define SDT_REACH 64                                 // search radius when building (must dwarf the kept candidates)

struct SdTable {
    int8_t off[SDT_NVEC][SDT_LEN][2];  // (dx, dy) relative to the edge's origin, best first
};

SD_FN int sdt_index(int ax, int ay) { return (ay + SDT_AMAX) * SDT_SIDE + (ax + SDT_AMAX); }

SD_FN bool sd_occupied(const SdGrid& g, int x, int y) {
    // branch-free: an out-of-image probe reads word 0 and is masked (a guarded load costs an exec-mask round trip per probe)
    const bool in = (unsigned)x < (unsigned)g.W && (unsigned)y < (unsigned)g.H;
    const int xi = in ? x : 0, yi = in ? y : 0;
    return in & (bool)((g.occ[SD_MUL(yi, g.wpr) + (xi >> 5)] >> (xi & 31)) & 1u);
}

SD_FN int32_t sd_orient(int ax, int ay, int bx, int by, int cx, int cy) {
    return SD_MUL(bx - ax, cy - ay) - SD_MUL(by - ay, cx - ax);
}

#if defined(__HIP_DEVICE_COMPILE__)
// Votes and broadcasts inside the group of lanes that shares a walk.  Control flow is uniform within a group, never
// between groups: a ballot taken under divergence simply lacks the other groups' bits, and a group only reads its own.
SD_FN unsigned long long sd_group_ballot(const SdGrid& g, bool pred) {
    const unsigned long long m = __ballot(pred);
    return g.nlanes == 64 ? m : (m >> g.gbase) & ((1ull << g.nlanes) - 1ull);
}
SD_FN int sd_group_read(const SdGrid& g, int v, int l) {  // v of the group's lane l (l: the same in all lanes of the group)
    return g.nlanes == 64 ? __builtin_amdgcn_readlane(v, l) : __shfl(v, g.gbase + l);
}
#endif

// True if no site can lie strictly on side dir of s->a because the bounding box of all sites does not reach there:
// then s->a is a hull edge.  Settles, without a sweep, the edges along the border of a cloud that was clipped to the
// image window (most hull edges).
SD_FN bool sd_side_is_empty(const SdGrid& g, int sx, int sy, int ax, int ay, int dir) {
    return sd_orient(sx, sy, ax, ay, g.bx0, g.by0) * dir <= 0 && sd_orient(sx, sy, ax, ay, g.bx1, g.by0) * dir <= 0 &&
           sd_orient(sx, sy, ax, ay, g.bx0, g.by1) * dir <= 0 && sd_orient(sx, sy, ax, ay, g.bx1, g.by1) * dir <= 0;
}

SD_FN bool sd_before(int ax, int ay, int bx, int by) { return ay < by || (ay == by && ax < bx); }

// > 0 iff d strictly inside the circle through a, b, c (a, b, c counter-clockwise).  Evaluated in float64, which is
// EXACT here: every intermediate is an integer below 2^48 (|coordinate difference| < 2^11), far inside the 53-bit
// significand -- and float64 runs at half the float32 rate on MI355X while 64-bit integer multiplies are emulated.
SD_FN double sd_incircle(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    const double adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    const double ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}

// Perturbed in-circle test, never a tie: true iff d is inside circle(a, b, c), a,b,c counter-clockwise.
SD_FN bool sd_inside(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    const double det = sd_incircle(ax, ay, bx, by, cx, cy, dx, dy);
    if (det != 0.0) return det > 0.0;
    // co-circular: the raster-first site carries the dominant perturbation
    bool a_first = sd_before(ax, ay, bx, by) && sd_before(ax, ay, cx, cy) && sd_before(ax, ay, dx, dy);
    bool b_first = sd_before(bx, by, ax, ay) && sd_before(bx, by, cx, cy) && sd_before(bx, by, dx, dy);
    bool c_first = sd_before(cx, cy, ax, ay) && sd_before(cx, cy, bx, by) && sd_before(cx, cy, dx, dy);
    if (a_first) return sd_orient(dx, dy, bx, by, cx, cy) > 0;
    if (b_first) return sd_orient(ax, ay, dx, dy, cx, cy) > 0;
    if (c_first) return sd_orient(ax, ay, bx, by, dx, dy) > 0;
    return false;  // d itself is first: raised out of the circle
}

// Candidate c beats the current apex p for the directed edge s->a on side dir (+1: left, -1: right).
SD_FN bool sd_better(int sx, int sy, int ax, int ay, int px, int py, int cx, int cy, int dir) {
    return dir > 0 ? sd_inside(sx, sy, ax, ay, px, py, cx, cy) : sd_inside(ax, ay, sx, sy, px, py, cx, cy);
}

struct SdCircle {
    double ox, oy, r2;
    double rpad;  // an upper bound of the radius, + 1 pixel (float32 square root, widened): used only to cut sweeps
};

SD_FN SdCircle sd_circle(int ax, int ay, int bx, int by, int cx, int cy) {
    double bxr = bx - ax, byr = by - ay, cxr = cx - ax, cyr = cy - ay;
    double d = 2.0 * (bxr * cyr - byr * cxr);
    double b2 = bxr * bxr + byr * byr, c2 = cxr * cxr + cyr * cyr;
    double ux = (cyr * b2 - byr * c2) / d, uy = (bxr * c2 - cxr * b2) / d;
    SdCircle c;
    c.ox = ax + ux;
    c.oy = ay + uy;
    c.r2 = ux * ux + uy * uy;
    c.rpad = (double)sqrtf((float)c.r2) * (1.0 + 1e-6) + 1.0;
    return c;
}

// Bits of row y restricted to columns [x0, x1] of word w (x0 <= x1, both inside the image).
SD_FN uint32_t sd_word_bits(const SdGrid& g, int y, int w, int x0, int x1) {
    uint32_t bits = g.occ[SD_MUL(y, g.wpr) + w];
    int lo = x0 - (w << 5), hi = x1 - (w << 5);
    if (lo > 0) bits &= 0xFFFFFFFFu << lo;
    if (hi < 31) bits &= 0xFFFFFFFFu >> (31 - hi);
    return bits;
}

SD_FN int sd_ctz(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)v) - 1;
#else
    return __builtin_ctz(v);
#endif
}

// Nearest site to s (on ties: any one of the nearest, chosen deterministically).  Returns false iff s is the only site.
// Lanes that share the search take rows lane, lane + nlanes, ... of each window and merge by a wave minimum.
SD_FN bool sd_nearest(const SdGrid& g, int sx, int sy, int* nx, int* ny) {
    int32_t best = INT32_MAX;
    int bx = -1, by = -1;
    int R = 2;
    for (;;) {
        int y0 = sy - R < 0 ? 0 : sy - R, y1 = sy + R > g.H - 1 ? g.H - 1 : sy + R;
        int x0 = sx - R < 0 ? 0 : sx - R, x1 = sx + R > g.W - 1 ? g.W - 1 : sx + R;
        for (int y = y0 + g.lane; y <= y1; y += g.nlanes) {
            if (g.rmax[y] < x0 || g.rmin[y] > x1) continue;
            for (int w = x0 >> 5; w <= (x1 >> 5); w++) {
                uint32_t bits = sd_word_bits(g, y, w, x0, x1);
                while (bits) {
                    int x = (w << 5) + sd_ctz(bits);
                    bits &= bits - 1;
                    if (x == sx && y == sy) continue;
                    const int32_t d2 = SD_MUL(x - sx, x - sx) + SD_MUL(y - sy, y - sy);
                    if (d2 < best) { best = d2; bx = x; by = y; }
                }
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (g.nlanes > 1) {
            int32_t m = best;
            for (int off = g.nlanes >> 1; off >= 1; off >>= 1) {
                const int32_t o = __shfl_xor(m, off);
                m = o < m ? o : m;
            }
            if (m != INT32_MAX) {  // the lowest lane holding the minimum
                const unsigned long long who = sd_group_ballot(g, best == m);
                const int l = (int)__ffsll((long long)who) - 1;
                bx = sd_group_read(g, bx, l);
                by = sd_group_read(g, by, l);
            }
            best = m;
        }
#endif
        if (best <= R * R) break;                                   // nothing outside the window is closer
        if (x0 == 0 && y0 == 0 && x1 == g.W - 1 && y1 == g.H - 1) break;    // whole image searched
        R *= 2;
    }
    *nx = bx;
    *ny = by;
    return bx >= 0;
}

#ifndef SD_COUNT
#define SD_COUNT(counter)
#endif
#ifndef SD_NOW
#define SD_NOW() 0
#define SD_LAP(slot, t) (void)(t)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Apex search by sweeps.  Candidates c on side dir of the edge s -> a are ranked by the parameter of their circle,
//     lambda(c) = (|c|^2 - a.c) / (2 |D|),   D = a x c   (coordinates relative to s),
// the signed position of the circle's centre along the edge's bisector: c1 lies inside circle(s, a, c2) iff
// lambda(c1) < lambda(c2).  Numerator and denominator are integers below 2^24 (coordinates below 2^11), exact in
// float32, so the float quotient is off by a few ulp at most: two candidates whose lambdas differ by more than
// SD_LAM_TOL (relative) are ordered by the floats alone, and only near-ties -- co-circular lattice points, mostly --
// go to the exact perturbed predicate (sd_better, float64).  The circle itself (centre = a/2 + dir lambda n,
// r^2 = |a|^2 (1/4 + lambda^2), n the left normal of a) costs a handful of float32 operations and is only used to cut
// the sweeps, always with margins that make the cut region a superset; circles too large for float32 cuts are not cut.
#define SD_LAM_TOL 2e-6f

#if defined(__HIP_DEVICE_COMPILE__)
#define SD_RCP(x) __builtin_amdgcn_rcpf(x)
#define SD_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define SD_RCP(x) (1.0f / (x))
#define SD_SQRT(x) sqrtf(x)
#endif

struct SdBest {
    int px, py;   // absolute; px < 0: none yet
    float lam;
    // the candidate's circle, for cutting sweeps (valid iff bounded): centre (absolute), padded radius, r^2
    float ox, oy, rpad, r2;
    bool bounded;
};

struct SdEdge {
    int sx, sy, ax, ay, dir;  // the edge (absolute) and the side searched
    int vx, vy;               // a - s
    // dir * orient(s, a, (x, y)) = P (y - sy) - Q (x - sx) >= 1  <=>  strictly on side dir of s -> a.
    // Small integers (< 2^11), so float32 products are exact; only the quotient is rounded, and the cut keeps a margin.
    float P, Q, invQ;
    float a2;                 // |a - s|^2
};

SD_FN SdEdge sd_edge(int sx, int sy, int ax, int ay, int dir) {
    SdEdge e;
    e.sx = sx; e.sy = sy; e.ax = ax; e.ay = ay; e.dir = dir;
    e.vx = ax - sx;
    e.vy = ay - sy;
    e.P = (float)(dir * e.vx);
    e.Q = (float)(dir * e.vy);
    e.invQ = e.Q != 0.f ? 1.0f / e.Q : 0.f;
    e.a2 = (float)(SD_MUL(e.vx, e.vx) + SD_MUL(e.vy, e.vy));
    return e;
}

SD_FN void sd_best_set(SdBest& b, const SdEdge& e, int x, int y, float lam) {
    b.px = x;
    b.py = y;
    b.lam = lam;
    const float r2 = e.a2 * (0.25f + lam * lam);
    b.r2 = r2;
    b.bounded = r2 < 6.0e7f;  // r < ~7700 px: float32 places the circle to ~1e-3 px
    const float t = (float)e.dir * lam;
    b.ox = (float)e.sx + 0.5f * (float)e.vx - t * (float)e.vy;
    b.oy = (float)e.sy + 0.5f * (float)e.vy + t * (float)e.vx;
    b.rpad = SD_SQRT(r2) * 1.00001f + 1.5f;
}

// lambda of the site (x, y), or false if it is not strictly on the searched side
SD_FN bool sd_lambda(const SdEdge& e, int x, int y, float* lam) {
    const int cx = x - e.sx, cy = y - e.sy;
    const int D = SD_MUL(e.vx, cy) - SD_MUL(e.vy, cx);
    if (e.dir > 0 ? D <= 0 : D >= 0) return false;
    const int N = SD_MUL(cx, cx) + SD_MUL(cy, cy) - (SD_MUL(e.vx, cx) + SD_MUL(e.vy, cy));
    *lam = (float)N * SD_RCP((float)(2 * (D < 0 ? -D : D)));
    return true;
}

// does candidate (x, y, lam) beat the current best?  floats when they are decisive, the exact predicate otherwise
SD_FN bool sd_beats(const SdEdge& e, const SdBest& b, int x, int y, float lam) {
    if (b.px < 0) return true;
    const float tol = SD_LAM_TOL * (fabsf(lam) + fabsf(b.lam)) + 1e-30f;
    if (lam < b.lam - tol) return true;
    if (lam > b.lam + tol) return false;
    if (x == b.px && y == b.py) return false;
    SD_COUNT(exact);
    return sd_better(e.sx, e.sy, e.ax, e.ay, b.px, b.py, x, y, e.dir);
}

// Scan rows ya, ya+step, ..., yb (step = +1 or -1), columns [xa, xb], for a better apex.  Rows are cut to their
// occupied extent and, before the bitmap is touched, to the open half-plane on the searched side and to the current
// candidate circle (supersets), so rows on the wrong side cost a few instructions and the work shrinks as the apex
// improves.  A sweep stops once it has left the circle in its direction of travel.
template <bool CUTS>
SD_FN void sd_scan_rows(const SdGrid& g, const SdEdge& e, int ya, int yb, int step, int xa, int xb, SdBest* best) {
    for (int y = ya + g.lane * step; step > 0 ? y <= yb : y >= yb; y += g.nlanes * step) {
        SD_COUNT(rows);
        int x0 = xa > g.rmin[y] ? xa : g.rmin[y];
        int x1 = xb < g.rmax[y] ? xb : g.rmax[y];
        if (x0 > x1) continue;
        const bool cut = CUTS && best->px >= 0 && best->bounded;
        float dy = 0.f;
        if (cut) {
            dy = (float)y - best->oy;
            if (dy > best->rpad || -dy > best->rpad) {
                if (step > 0 ? dy > 0.f : dy < 0.f) break;  // left the circle for good
                continue;
            }
        }
        {
            const float T = e.P * (float)(y - e.sy) - 1.0f;  // exact
            if (e.Q == 0.f) {
                if (T < 0.f) continue;
            } else {
                const float t = T * e.invQ;  // Q u <= T  with u = x - sx ; |t| < 2^22, rounding error < 1
                if (e.Q > 0.f) {
                    const float hi = floorf(t) + 2.0f + (float)e.sx;
                    if (hi < (float)x1) x1 = hi < -1.0f ? -1 : (int)hi;
                } else {
                    const float lo = ceilf(t) - 2.0f + (float)e.sx;
                    if (lo > (float)x0) x0 = lo > 1e9f ? g.W : (int)lo;
                }
                if (x0 > x1) continue;
            }
            if (cut) {
                const float h2 = best->r2 - dy * dy;
                // r^2 - dy^2 cancels: widen by its round-off (a few ulp of r^2) before the square root
                const float half = SD_SQRT((h2 > 0.f ? h2 : 0.f) + 2e-6f * best->r2) * 1.00001f + 1.5f;
                const float lo = floorf(best->ox - half), hi = ceilf(best->ox + half);
                if (lo > (float)x0) x0 = (int)lo;
                if (hi < (float)x1) x1 = (int)hi;
                if (x0 > x1) continue;
            }
        }
        for (int w = x0 >> 5; w <= (x1 >> 5); w++) {
            uint32_t bits = sd_word_bits(g, y, w, x0, x1);
            while (bits) {
                const int x = (w << 5) + sd_ctz(bits);
                bits &= bits - 1;
                SD_COUNT(bits);
                float lam;
                if (!sd_lambda(e, x, y, &lam)) continue;  // wrong side, collinear, or s / a themselves
                if (sd_beats(e, *best, x, y, lam)) {
                    if (CUTS) {
                        sd_best_set(*best, e, x, y, lam);
                    } else {  // small window: the circle is only needed after the merge (sd_share_best sets it)
                        best->px = x;
                        best->py = y;
                        best->lam = lam;
                    }
                }
            }
        }
    }
}

// Merge the candidates of the lanes that share a sweep, after which every lane holds the same best apex: a float
// minimum over the wave finds the contenders (lambda within the tolerance of the minimum -- the true best is among them),
// which are then folded with the exact order (one contender in the common case).
SD_FN void sd_share_best(const SdGrid& g, const SdEdge& e, SdBest* best, int* shared_x, int* shared_y) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (g.nlanes > 1) {
        // nothing to merge if no lane improved on the candidate all lanes agreed on last time
        const unsigned long long changed = sd_group_ballot(g, best->px >= 0 && (best->px != *shared_x || best->py != *shared_y));
        if (changed == 0ull) return;
        // contenders: the lanes that changed (each of their candidates beats the one shared so far); if many did, a
        // float minimum over the wave thins them out first
        unsigned long long cont = changed;
        if (__popcll(changed) > 4) {
            float m = best->px >= 0 ? best->lam : 3.0e38f;
            for (int off = g.nlanes >> 1; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off));
            const float tol = 2.f * SD_LAM_TOL * fabsf(m) + 1e-30f;
            cont = sd_group_ballot(g, best->px >= 0 && best->lam <= m + 2.f * tol);
        }
        SdBest w;
        w.px = -1; w.py = -1; w.lam = 0.f; w.ox = w.oy = w.rpad = w.r2 = 0.f; w.bounded = false;
        while (cont) {
            const int l = (int)__ffsll((long long)cont) - 1;
            cont &= cont - 1ull;
            const int ox = sd_group_read(g, best->px, l), oy = sd_group_read(g, best->py, l);
            const float ol = __uint_as_float((unsigned)sd_group_read(g, (int)__float_as_uint(best->lam), l));
            if (sd_beats(e, w, ox, oy, ol)) { w.px = ox; w.py = oy; w.lam = ol; }
        }
        // (`changed` is never empty here, and the minimum's own lane is a contender in exact arithmetic; should the float thinning
        //  ever drop every lane -- a NaN lambda would --, EVERY lane goes back to the candidate shared so far, so that the function's
        //  post-condition -- all lanes hold the same best apex -- survives: a lane that kept its private candidate would cut its rows
        //  with a circle of its own in the next sweep)
        if (w.px < 0) {
            float lam;
            if (*shared_x >= 0 && sd_lambda(e, *shared_x, *shared_y, &lam)) sd_best_set(*best, e, *shared_x, *shared_y, lam);
            else { best->px = -1; best->py = -1; }
            return;
        }
        sd_best_set(*best, e, w.px, w.py, w.lam);
        *shared_x = w.px;
        *shared_y = w.py;
    }
#else
    (void)g; (void)e; (void)best; (void)shared_x; (void)shared_y;
#endif
}

// First window of a sweeping query: the edge's bounding box widened by this many pixels.  Round 4, densify per 4096 renders
// (bit-identical images): 6 -> 14.15-14.35 ms, 9 -> 13.9-14.1, 12 -> 14.05-14.16, 16 -> 14.3; with the circle cuts of the second sweep
// applied in the windows too 16.0 (their square roots cost more than the candidates they save in a small window), and the circle
// sweep WITHOUT its per-row chord cut 18.2: what a sweep costs is the candidates it evaluates.
#ifndef SD_WINDOW_MARGIN
#define SD_WINDOW_MARGIN 9
#endif

// Apex of the Delaunay triangle on side dir of the Delaunay edge s->a.  Returns false iff there is no site
// strictly on that side, i.e. s->a is a hull edge.
SD_FN bool sd_apex(const SdGrid& g, int sx, int sy, int ax, int ay, int dir, int* outx, int* outy) {
    SD_COUNT(apex);
    long long lap = SD_NOW();
    // 0. short edge: the pre-sorted candidate table answers with bitmap probes alone.  (Side -1 of s->a is side +1 of
    //    a->s, so the table is entered with the edge reversed.)  First the SDT_LEAN_LEN entries the lean walk knows -- one probe
    //    per lane --, the rest of the table only after the cheaper answers (bounding box, triangle cache) have failed.
    const int tox = dir > 0 ? sx : ax, toy = dir > 0 ? sy : ay;
    const int tvx = dir > 0 ? ax - sx : sx - ax, tvy = dir > 0 ? ay - sy : sy - ay;
    const bool tabled = g.tab != nullptr && tvx >= -SDT_AMAX && tvx <= SDT_AMAX && tvy >= -SDT_AMAX && tvy <= SDT_AMAX;
    const int8_t* row = tabled ? g.tab + sdt_index(tvx, tvy) * (SDT_LEN * 2) : nullptr;
    if (tabled) {
        int hit = -1;
#if defined(__HIP_DEVICE_COMPILE__)
        if (g.nlanes > 1) {  // one candidate per lane and round, the lowest occupied entry wins
            for (int k0 = 0; k0 < SDT_LEAN_LEN && hit < 0; k0 += g.nlanes) {
                const int k = k0 + g.lane;
                const bool b = k < SDT_LEAN_LEN && sd_occupied(g, tox + row[2 * (k & (SDT_LEN - 1))], toy + row[2 * (k & (SDT_LEN - 1)) + 1]);
                const unsigned long long m = sd_group_ballot(g, b);
                if (m) hit = k0 + (int)__ffsll((long long)m) - 1;
            }
        } else
#endif
        {
            for (int k = 0; k < SDT_LEAN_LEN && hit < 0; k++)
                if (sd_occupied(g, tox + row[2 * k], toy + row[2 * k + 1])) hit = k;
        }
        SD_LAP(table, lap);
        if (hit >= 0) {
            SD_COUNT(apex_table);
            *outx = tox + row[2 * hit];
            *outy = toy + row[2 * hit + 1];
            return true;
        }
    }
    if (sd_side_is_empty(g, sx, sy, ax, ay, dir)) return false;
    // the side `dir` of s -> a is the left of (u -> v) = (s -> a) or (a -> s)
    const int ux = dir > 0 ? sx : ax, uy = dir > 0 ? sy : ay, vx = dir > 0 ? ax : sx, vy = dir > 0 ? ay : sy;
    {
        const int hit = sd_cache_lookup(g, ux, uy, vx, vy, outx, outy);
        if (hit) {
            SD_COUNT(apex_cached);
            return hit == 1;
        }
    }
    // 0b. the REST of the table (entries SDT_LEAN_LEN .. SDT_LEN - 1: the circle-growth order out to ~20 pixels).  An apex found
    //     here needs no sweep at all -- the table order is exact -- and a sweeping query costs two sweeps and their merges.  All
    //     rounds' table entries are fetched before any is used (one trip to the table, one to the bitmap), then the first
    //     occupied entry in table order wins.
    if (tabled && SDT_LEN > SDT_LEAN_LEN) {
        int hit = -1;
#if defined(__HIP_DEVICE_COMPILE__)
        if (g.nlanes > 1) {
            constexpr int NR = (SDT_LEN - SDT_LEAN_LEN + 63) / 64;   // rounds of one candidate per lane (nlanes = 64 here; smaller groups loop)
            if (g.nlanes == 64) {
                int cxs[NR], cys[NR];
                bool occb[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const int k = SDT_LEAN_LEN + 64 * r + g.lane;
                    cxs[r] = k < SDT_LEN ? row[2 * (k & (SDT_LEN - 1))] : 127;
                    cys[r] = k < SDT_LEN ? row[2 * (k & (SDT_LEN - 1)) + 1] : 127;
                }
#pragma unroll
                for (int r = 0; r < NR; r++) occb[r] = (SDT_LEAN_LEN + 64 * r + g.lane) < SDT_LEN && sd_occupied(g, tox + cxs[r], toy + cys[r]);
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const unsigned long long m = sd_group_ballot(g, occb[r]);
                    if (m != 0ull && hit < 0) hit = SDT_LEAN_LEN + 64 * r + (int)__ffsll((long long)m) - 1;
                }
            } else {
                for (int k0 = SDT_LEAN_LEN; k0 < SDT_LEN && hit < 0; k0 += g.nlanes) {
                    const int k = k0 + g.lane;
                    const bool b = k < SDT_LEN && sd_occupied(g, tox + row[2 * (k & (SDT_LEN - 1))], toy + row[2 * (k & (SDT_LEN - 1)) + 1]);
                    const unsigned long long m = sd_group_ballot(g, b);
                    if (m) hit = k0 + (int)__ffsll((long long)m) - 1;
                }
            }
        } else
#endif
        {
            for (int k = SDT_LEAN_LEN; k < SDT_LEN && hit < 0; k++)
                if (sd_occupied(g, tox + row[2 * k], toy + row[2 * k + 1])) hit = k;
        }
        SD_LAP(table, lap);
        if (hit >= 0) {
            SD_COUNT(apex_table);
            *outx = tox + row[2 * hit];
            *outy = toy + row[2 * hit + 1];
            sd_cache_insert(g, ux, uy, vx, vy, *outx, *outy);
            return true;
        }
    }
    const SdEdge edge = sd_edge(sx, sy, ax, ay, dir);
    SdBest best;
    best.px = best.py = -1;
    best.lam = 0.f;
    best.ox = best.oy = best.rpad = best.r2 = 0.f;
    best.bounded = false;
    int shx = -1, shy = -1;  // the candidate all sharing lanes last agreed on
    // 1. a small window around the edge finds the apex in dense regions; while it finds nothing it is widened
    //    threefold (clipped to the bounding box of the sites).  Searching outwards keeps the first candidates few:
    //    the previous, smaller window was empty, and every candidate found shrinks the circle for the others.
    //    A window that covers the whole bounding box and is still empty proves s->a a hull edge.
    int M = SD_WINDOW_MARGIN;
    int wy0, wy1, wx0, wx1;
    for (;;) {
        wy0 = (sy < ay ? sy : ay) - M; wy1 = (sy > ay ? sy : ay) + M;
        wx0 = (sx < ax ? sx : ax) - M; wx1 = (sx > ax ? sx : ax) + M;
        if (wy0 < g.by0) wy0 = g.by0;
        if (wx0 < g.bx0) wx0 = g.bx0;
        if (wy1 > g.by1) wy1 = g.by1;
        if (wx1 > g.bx1) wx1 = g.bx1;
        sd_scan_rows<false>(g, edge, wy0, wy1, 1, wx0, wx1, &best);
        SD_LAP(window, lap);
        sd_share_best(g, edge, &best, &shx, &shy);
        SD_LAP(share, lap);
        if (best.px >= 0) break;
        if (wy0 <= g.by0 && wx0 <= g.bx0 && wy1 >= g.by1 && wx1 >= g.bx1) {
            SD_LAP(far, lap);
            sd_cache_insert_hull(g, ux, uy, vx, vy);
            return false;
        }
        SD_COUNT(apex_far);
        M = 3 * M + 2;
    }
    if (g.nlanes == 1) sd_best_set(best, edge, best.px, best.py, best.lam);  // (a merge sets the circle itself)
    SD_LAP(window, lap);
    {
        // 2. whatever part of the candidate's circle sticks out of the window (and holds sites) is swept too
        int cy0 = g.by0, cy1 = g.by1, cx0 = g.bx0, cx1 = g.bx1;
        if (best.bounded) {
            const float fy0 = floorf(best.oy - best.rpad), fy1 = ceilf(best.oy + best.rpad);
            const float fx0 = floorf(best.ox - best.rpad), fx1 = ceilf(best.ox + best.rpad);
            if (fy0 > (float)cy0) cy0 = (int)fy0;
            if (fx0 > (float)cx0) cx0 = (int)fx0;
            if (fy1 < (float)cy1) cy1 = (int)fy1;
            if (fx1 < (float)cx1) cx1 = (int)fx1;
        }
        if (best.lam <= 0.f) {
            // The centre lies on the far side of the edge (or on it): what remains to search is the MINOR segment of
            // the disc, every point of which is within the sagitta h of the chord s-a and projects onto it -- so it
            // lies in the chord's bounding box widened by h.  For the slivers along a straight outline (three nearly
            // collinear sites, a circle hundreds of pixels across) that is a few rows instead of the whole image.
            // h = r - d = (|a|/2)^2 / (r + d), d = |lambda| |a| the centre's distance from the chord (no cancellation).
            // (Better candidates found during the sweep have their segment on this side INSIDE this one: the circles
            // through s and a are nested on either side of the chord, ordered by lambda.)
            const float h = 0.25f * edge.a2 / (SD_SQRT(best.r2) + fabsf(best.lam) * SD_SQRT(edge.a2)) * 1.0001f + 2.0f;
            const float ly0 = floorf((float)(sy < ay ? sy : ay) - h), ly1 = ceilf((float)(sy > ay ? sy : ay) + h);
            const float lx0 = floorf((float)(sx < ax ? sx : ax) - h), lx1 = ceilf((float)(sx > ax ? sx : ax) + h);
            if (ly0 > (float)cy0) cy0 = (int)ly0;
            if (lx0 > (float)cx0) cx0 = (int)lx0;
            if (ly1 < (float)cy1) cy1 = (int)ly1;
            if (lx1 < (float)cx1) cx1 = (int)lx1;
        }
        if (cy1 > wy1 || cy0 < wy0 || cx0 < wx0 || cx1 > wx1) {
            SD_COUNT(apex_slow);
            // one sweep over the circle's rows: what the window already covered is cut away by the circle, cheaply
            sd_scan_rows<true>(g, edge, cy0, cy1, 1, cx0, cx1, &best);
            SD_LAP(slow, lap);
            sd_share_best(g, edge, &best, &shx, &shy);
            SD_LAP(share, lap);
        }
        SD_LAP(slow, lap);
    }
    *outx = best.px;
    *outy = best.py;
    sd_cache_insert(g, ux, uy, vx, vy, best.px, best.py);
    return true;
}

#define SD_MAX_DEGREE 8192  // safety bound on the wrap loop; a lattice site cannot have more neighbours than this

// Walk the Delaunay star of site s and hand every triangle whose raster-first vertex is s to `emit`
// (so each triangle of the triangulation is emitted exactly once over all sites), counter-clockwise.
// Returns the number of wrap steps, or -1 if the safety bound was hit.
//
// Only triangles whose other two vertices FOLLOW s in raster order are emitted, i.e. neighbours at angles [0, pi)
// counter-clockwise from +x.  Three modes, ONE loop (and one inlined copy of sd_apex and of the emit functor):
//   HALF  the pixel to the right is a site -- the first of those neighbours, adjacent pixels always being Delaunay
//         neighbours: walk counter-clockwise from it and stop at the first neighbour that precedes s, or at the hull;
//   CCW   otherwise: counter-clockwise from some neighbour n0 (the nearest site) until the star closes at n0; if the hull
//         comes first, s is a hull vertex and the fan is finished  CW  (clockwise) from n0 to the hull on the other side.
// `fresh` = false takes the walk up where a lean walk (star_local.h) gave up: at the edge s -> s + (rax, ray) it could not
// answer, in its direction, with its first neighbour n0 = s + (rn0x, rn0y) or as a half walk; what that walk emitted before
// is not emitted again.
enum { SD_MODE_HALF = 0, SD_MODE_CCW = 1, SD_MODE_CW = 2 };
template <class Emit>
SD_FN int sd_walk(const SdGrid& g, int sx, int sy, bool fresh, int rax, int ray, int rdir, bool rhalf, int rn0x, int rn0y, Emit& emit) {
    int steps = 0;
    int mode, ax, ay, n0x, n0y;
    if (!fresh) {
        mode = rhalf ? SD_MODE_HALF : (rdir > 0 ? SD_MODE_CCW : SD_MODE_CW);
        ax = sx + rax; ay = sy + ray; n0x = sx + rn0x; n0y = sy + rn0y;
    } else if (sx + 1 < g.W && ((g.occ[SD_MUL(sy, g.wpr) + ((sx + 1) >> 5)] >> ((sx + 1) & 31)) & 1u)) {
        mode = SD_MODE_HALF;
        ax = n0x = sx + 1; ay = n0y = sy;
    } else {
        long long lap_n = SD_NOW();
        if (!sd_nearest(g, sx, sy, &n0x, &n0y)) return 0;
        SD_LAP(nearest, lap_n);
        mode = SD_MODE_CCW;
        ax = n0x; ay = n0y;
    }
    const bool half = mode == SD_MODE_HALF;
    for (;;) {
        int px, py;
        if (!sd_apex(g, sx, sy, ax, ay, mode == SD_MODE_CW ? -1 : +1, &px, &py)) {
            if (mode != SD_MODE_CCW) break;
            mode = SD_MODE_CW;  // hull vertex: back to the first neighbour, the other way round
            ax = n0x;
            ay = n0y;
            continue;
        }
        if (half && !sd_before(sx, sy, px, py)) break;
        if (half || (sd_before(sx, sy, ax, ay) && sd_before(sx, sy, px, py))) {
            const bool cw = mode == SD_MODE_CW;  // (counter-clockwise vertex order either way)
            long long lap_e = SD_NOW();
            emit(sx, sy, cw ? px : ax, cw ? py : ay, cw ? ax : px, cw ? ay : py);
            SD_LAP(e1_total, lap_e);
        }
        if (++steps > SD_MAX_DEGREE) return -1;
        ax = px;
        ay = py;
        if (mode == SD_MODE_CCW && ax == n0x && ay == n0y) break;  // closed
    }
    return half ? steps + 1 : steps;
}

template <class Emit>
SD_FN int sd_star(const SdGrid& g, int sx, int sy, Emit& emit) {
    return sd_walk(g, sx, sy, true, 0, 0, 1, false, 0, 0, emit);
}

template <class Emit>
SD_FN int sd_star_resume(const SdGrid& g, int sx, int sy, int rax, int ray, int dir, bool half, int rn0x, int rn0y, Emit& emit) {
    return sd_walk(g, sx, sy, false, rax, ray, dir, half, rn0x, rn0y, emit);
}
