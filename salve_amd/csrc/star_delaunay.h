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

struct SdGrid {
    int H, W, wpr;        // image height, width, 32-bit words per bitmap row
    const uint32_t* occ;  // [H][wpr] occupancy bits, bit (x & 31) of word x >> 5
    const int16_t* rmin;  // [H] smallest occupied x of the row, or W if the row is empty
    const int16_t* rmax;  // [H] largest occupied x of the row, or -1
};

SD_FN int32_t sd_orient(int ax, int ay, int bx, int by, int cx, int cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}

SD_FN bool sd_before(int ax, int ay, int bx, int by) { return ay < by || (ay == by && ax < bx); }

// > 0 iff d strictly inside the circle through a, b, c (a, b, c counter-clockwise); exact.
SD_FN int64_t sd_incircle(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    int64_t adx = ax - dx, ady = ay - dy, bdx = bx - dx, bdy = by - dy, cdx = cx - dx, cdy = cy - dy;
    int64_t ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}

// Perturbed in-circle test, never a tie: true iff d is inside circle(a, b, c), a,b,c counter-clockwise.
SD_FN bool sd_inside(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    int64_t det = sd_incircle(ax, ay, bx, by, cx, cy, dx, dy);
    if (det != 0) return det > 0;
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
    return c;
}

// Bits of row y restricted to columns [x0, x1] of word w (x0 <= x1, both inside the image).
SD_FN uint32_t sd_word_bits(const SdGrid& g, int y, int w, int x0, int x1) {
    uint32_t bits = g.occ[y * g.wpr + w];
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

// Nearest site to s (any one of the nearest on ties).  Returns false iff s is the only site.
SD_FN bool sd_nearest(const SdGrid& g, int sx, int sy, int* nx, int* ny) {
    int64_t best = INT64_MAX;
    int bx = -1, by = -1;
    int R = 2;
    for (;;) {
        int y0 = sy - R < 0 ? 0 : sy - R, y1 = sy + R > g.H - 1 ? g.H - 1 : sy + R;
        int x0 = sx - R < 0 ? 0 : sx - R, x1 = sx + R > g.W - 1 ? g.W - 1 : sx + R;
        for (int y = y0; y <= y1; y++) {
            if (g.rmax[y] < x0 || g.rmin[y] > x1) continue;
            for (int w = x0 >> 5; w <= (x1 >> 5); w++) {
                uint32_t bits = sd_word_bits(g, y, w, x0, x1);
                while (bits) {
                    int x = (w << 5) + sd_ctz(bits);
                    bits &= bits - 1;
                    if (x == sx && y == sy) continue;
                    int64_t d2 = (int64_t)(x - sx) * (x - sx) + (int64_t)(y - sy) * (y - sy);
                    if (d2 < best) { best = d2; bx = x; by = y; }
                }
            }
        }
        if (best <= (int64_t)R * R) break;                                   // nothing outside the window is closer
        if (x0 == 0 && y0 == 0 && x1 == g.W - 1 && y1 == g.H - 1) break;    // whole image searched
        R *= 2;
    }
    *nx = bx;
    *ny = by;
    return bx >= 0;
}

// Scan rows [ya, yb] for a better apex of edge s->a on side dir.  (px,py) < 0 means "none yet".
// The x-range of each row is cut to the current circle (a superset with a one-pixel margin), so the work shrinks
// as the apex improves; rows are cut to the occupied extent.
SD_FN void sd_scan_rows(const SdGrid& g, int sx, int sy, int ax, int ay, int dir, int ya, int yb, int xa, int xb,
                        int* px, int* py, SdCircle* circ) {
    for (int y = ya; y <= yb; y++) {
        int x0 = xa > g.rmin[y] ? xa : g.rmin[y];
        int x1 = xb < g.rmax[y] ? xb : g.rmax[y];
        if (*px >= 0) {
            double dy = y - circ->oy;
            double h2 = circ->r2 - dy * dy;
            double r1 = sqrt(circ->r2) + 1.0;
            if (dy > r1 || -dy > r1) continue;
            // h2 carries a round-off of a few ulp of r^2 (r can reach 2.5e8 px for sliver triangles): widen by it
            double half = sqrt((h2 > 0 ? h2 : 0.0) + 4e-15 * circ->r2) + 1.0;
            double lo = floor(circ->ox - half), hi = ceil(circ->ox + half);
            if (lo > x0) x0 = lo > 1e9 ? g.W : (int)lo;
            if (hi < x1) x1 = hi < -1e9 ? -1 : (int)hi;
        }
        if (x0 > x1) continue;
        for (int w = x0 >> 5; w <= (x1 >> 5); w++) {
            uint32_t bits = sd_word_bits(g, y, w, x0, x1);
            while (bits) {
                int x = (w << 5) + sd_ctz(bits);
                bits &= bits - 1;
                int32_t o = sd_orient(sx, sy, ax, ay, x, y);
                if (dir > 0 ? o <= 0 : o >= 0) continue;  // wrong side, collinear, or s / a themselves
                if (*px < 0 || sd_better(sx, sy, ax, ay, *px, *py, x, y, dir)) {
                    *px = x;
                    *py = y;
                    *circ = sd_circle(sx, sy, ax, ay, x, y);
                }
            }
        }
    }
}

// Apex of the Delaunay triangle on side dir of the Delaunay edge s->a.  Returns false iff there is no site
// strictly on that side, i.e. s->a is a hull edge.
SD_FN bool sd_apex(const SdGrid& g, int sx, int sy, int ax, int ay, int dir, int* outx, int* outy) {
    int px = -1, py = -1;
    SdCircle circ = {0, 0, 0};
    // 1. a small window around the edge finds the apex in dense regions
    const int M = 3;
    int wy0 = (sy < ay ? sy : ay) - M, wy1 = (sy > ay ? sy : ay) + M;
    int wx0 = (sx < ax ? sx : ax) - M, wx1 = (sx > ax ? sx : ax) + M;
    if (wy0 < 0) wy0 = 0;
    if (wx0 < 0) wx0 = 0;
    if (wy1 > g.H - 1) wy1 = g.H - 1;
    if (wx1 > g.W - 1) wx1 = g.W - 1;
    sd_scan_rows(g, sx, sy, ax, ay, dir, wy0, wy1, wx0, wx1, &px, &py, &circ);
    if (px >= 0) {
        // 2. the candidate's circle may stick out of the window: sweep what is left of its bounding box
        double r = sqrt(circ.r2) + 2.0;
        double fy0 = circ.oy - r, fy1 = circ.oy + r, fx0 = circ.ox - r, fx1 = circ.ox + r;
        int ya = fy0 <= 0 ? 0 : (int)fy0, yb = fy1 >= g.H - 1 ? g.H - 1 : (int)fy1 + 1;
        int xa = fx0 <= 0 ? 0 : (int)fx0, xb = fx1 >= g.W - 1 ? g.W - 1 : (int)fx1 + 1;
        if (yb > g.H - 1) yb = g.H - 1;
        if (xb > g.W - 1) xb = g.W - 1;
        if (ya < wy0) sd_scan_rows(g, sx, sy, ax, ay, dir, ya, wy0 - 1, xa, xb, &px, &py, &circ);
        if (yb > wy1) sd_scan_rows(g, sx, sy, ax, ay, dir, wy1 + 1, yb, xa, xb, &px, &py, &circ);
        if (xa < wx0) sd_scan_rows(g, sx, sy, ax, ay, dir, wy0, wy1, xa, wx0 - 1, &px, &py, &circ);
        if (xb > wx1) sd_scan_rows(g, sx, sy, ax, ay, dir, wy0, wy1, wx1 + 1, xb, &px, &py, &circ);
    } else {
        // 3. nothing near the edge: sweep the whole image (rows are cut to their occupied extent, and to the
        //    circle as soon as a first candidate turns up).  An empty result means s->a is a hull edge.
        sd_scan_rows(g, sx, sy, ax, ay, dir, 0, g.H - 1, 0, g.W - 1, &px, &py, &circ);
    }
    *outx = px;
    *outy = py;
    return px >= 0;
}

#define SD_MAX_DEGREE 8192  // safety bound on the wrap loop; a lattice site cannot have more neighbours than this

// Walk the Delaunay star of site s and hand every triangle whose raster-first vertex is s to `emit`
// (so each triangle of the triangulation is emitted exactly once over all sites), counter-clockwise.
// Returns the number of wrap steps, or -1 if the safety bound was hit.
template <class Emit>
SD_FN int sd_star(const SdGrid& g, int sx, int sy, Emit& emit) {
    int n0x, n0y;
    if (!sd_nearest(g, sx, sy, &n0x, &n0y)) return 0;
    int steps = 0;
    int ax = n0x, ay = n0y;
    bool closed = false;
    for (;;) {  // counter-clockwise from the nearest neighbour
        int px, py;
        if (!sd_apex(g, sx, sy, ax, ay, +1, &px, &py)) break;
        if (sd_before(sx, sy, ax, ay) && sd_before(sx, sy, px, py)) emit(sx, sy, ax, ay, px, py);
        if (++steps > SD_MAX_DEGREE) return -1;
        ax = px;
        ay = py;
        if (ax == n0x && ay == n0y) { closed = true; break; }
    }
    if (!closed) {  // s is a hull vertex: finish the fan clockwise from the nearest neighbour
        ax = n0x;
        ay = n0y;
        for (;;) {
            int px, py;
            if (!sd_apex(g, sx, sy, ax, ay, -1, &px, &py)) break;
            if (sd_before(sx, sy, ax, ay) && sd_before(sx, sy, px, py)) emit(sx, sy, px, py, ax, ay);
            if (++steps > SD_MAX_DEGREE) return -1;
            ax = px;
            ay = py;
        }
    }
    return steps;
}
