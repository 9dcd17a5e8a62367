// star_local.h -- the Delaunay star walk of star_delaunay.h as a per-lane STATE MACHINE over a 32 x 15 pixel window
// of the LDS bitmap around the site, written for SIMT execution on 64-wide wavefronts.
//
// Why: with one site per lane and the textbook nested loops (steps -> rows -> words -> bits) a wavefront pays, at
// every nesting level, for its slowest lane; measured lane utilisation of that form was 15 %.  Here every lane
// runs ONE flat loop whose iteration is "advance to the next window row" or "test one candidate"; a lane
// that finishes a site immediately pulls the next one, so lanes never wait for each other and the only cost of
// divergence is that both (short) bodies are issued.
//
// Exactness: inside the window all coordinates are relative to the site and bounded by 16, so the orientation and
// in-circle determinants are integers below 2^24 and are evaluated EXACTLY in float32 (full-rate VALU).  A query is
// accepted only if its final circle lies inside the window (then every site that could matter was examined with the
// exact predicates); otherwise the whole site is handed to the general algorithm (sd_star), as are hull sites.
// The symbolic perturbation (raster order of the sites) is the one of star_delaunay.h.
#pragma once
#include "star_delaunay.h"
#include "star_table.h"

// Square roots and reciprocals below only size search masks and the (conservative) acceptance test, always with a
// margin that dwarfs one ulp, so the device uses the single-instruction approximations.
#if defined(__HIP_DEVICE_COMPILE__)
#define SDL_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define SDL_RCP(x) __builtin_amdgcn_rcpf(x)
#else
#define SDL_SQRT(x) sqrtf(x)
#define SDL_RCP(x) (1.0f / (x))
#endif

#define SDL_ROWS 15      // window rows: sy-7 .. sy+7
#define SDL_HALF 7
#define SDL_XLO (-16)    // window columns: sx-16 .. sx+15 (bit = dx + 16)
#define SDL_XHI 15
#define SDL_NONE 0x7FFF

#ifndef SDL_HARD
#define SDL_HARD(reason) SDL_SITE_HARD   // a host build may count the reasons
#endif
enum { SDL_CONTINUE = 0, SDL_SITE_DONE = 1, SDL_SITE_HARD = 2 };
enum { SDL_MODE_NEAREST = 0, SDL_MODE_APEX = 1 };

struct SdLocal {
    int sx, sy;
    int n0x, n0y;      // first neighbour (relative)
    int ax, ay;        // current edge s -> a (relative)
    int px, py;        // best apex so far (px == SDL_NONE: none)
    float ux, uy, r2;  // circle through s, a, p (relative to s); for MODE_NEAREST: centre s, r2 = best distance^2
    int mode, k, m, row, deg, stage;
    float hx, hy, hr2, inv_ay;  // stage 0 looks only inside this disc around the edge; a second sweep follows if that was not enough
    uint32_t bits;
    bool upDone, dnDone;
    int dir;    // +1: walking counter-clockwise (apex left of s -> a); -1: clockwise, after the hull was met (table queries only)
    bool cert;  // the apex came from the table (star_table.h): exact and complete by construction, no circle test needed
    bool half;  // the walk started at the neighbour (+1, 0): stop at the first neighbour that precedes s in raster order
};

// 32 bits of bitmap row y starting at column x0 (may be negative / beyond the image: zeros)
SD_FN uint32_t sdl_row32(const SdGrid& g, int y, int x0) {
    if (y < 0 || y >= g.H) return 0u;
    const int w0 = x0 >> 5, sh = x0 & 31;  // arithmetic shift: floor
    const uint32_t lo = (w0 >= 0 && w0 < g.wpr) ? g.occ[y * g.wpr + w0] : 0u;
    const uint32_t hi = (w0 + 1 >= 0 && w0 + 1 < g.wpr) ? g.occ[y * g.wpr + w0 + 1] : 0u;
    return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
}

SD_FN uint32_t sdl_range_mask(int xl, int xr) {  // bits for relative columns xl..xr, clipped to the window
    if (xl < SDL_XLO) xl = SDL_XLO;
    if (xr > SDL_XHI) xr = SDL_XHI;
    if (xl > xr) return 0u;
    const int lo = xl - SDL_XLO, hi = xr - SDL_XLO;
    return (0xFFFFFFFFu << lo) & (0xFFFFFFFFu >> (31 - hi));
}

SD_FN void sdl_restart_scan(SdLocal& s) {
    s.k = -1;
    s.bits = 0;
    s.upDone = s.dnDone = false;
}

SD_FN bool sdl_bit(const SdGrid& g, int x, int y) {
    return x >= 0 && x < g.W && y >= 0 && y < g.H && ((g.occ[y * g.wpr + (x >> 5)] >> (x & 31)) & 1u);
}

SD_FN void sdl_set_apex(SdLocal& s, int cx, int cy);
SD_FN bool sdl_inside(int ax, int ay, int px, int py, int cx, int cy);

SD_FN void sdl_start_query(SdLocal& s, const SdGrid& g, int mode) {
    s.mode = mode;
    s.px = SDL_NONE;
    s.py = 0;
    s.stage = 0;
    s.cert = false;
    sdl_restart_scan(s);
    if (mode == SDL_MODE_APEX) {
        s.m = (s.ay + (s.ay >= 0 ? 1 : 0)) >> 1;  // a row next to the middle of the edge
        s.hx = 0.5f * (float)s.ax;
        s.hy = 0.5f * (float)s.ay;
        const float hr = 0.5f * SDL_SQRT((float)(s.ax * s.ax + s.ay * s.ay)) + 2.5f;
        s.hr2 = hr * hr;
        s.inv_ay = s.ay != 0 ? SDL_RCP((float)s.ay) : 0.f;
        // Short edges (nearly all edges where the image is densely covered): the candidates come pre-sorted from the
        // table, the first occupied one IS the apex.  Four entries per table read, four independent bitmap probes.
        if (g.tab != nullptr && s.ax >= -SDT_AMAX && s.ax <= SDT_AMAX && s.ay >= -SDT_AMAX && s.ay <= SDT_AMAX) {
            // the right of s -> a is the left of a -> s: clockwise queries enter the table with the edge reversed
            const int ox = s.dir > 0 ? 0 : s.ax, oy = s.dir > 0 ? 0 : s.ay;
            const int vx = s.dir > 0 ? s.ax : -s.ax, vy = s.dir > 0 ? s.ay : -s.ay;
            const uint32_t* row = (const uint32_t*)(g.tab + sdt_index(vx, vy) * (SDT_LEN * 2));
            const int bx = s.sx + ox, by = s.sy + oy;
            for (int k = 0; k < SDT_LEN / 4; k++) {
                const uint32_t e0 = row[2 * k], e1 = row[2 * k + 1];
                const int x0 = (int8_t)(e0 & 0xFF), y0 = (int8_t)((e0 >> 8) & 0xFF), x1 = (int8_t)((e0 >> 16) & 0xFF), y1 = (int8_t)(e0 >> 24);
                const int x2 = (int8_t)(e1 & 0xFF), y2 = (int8_t)((e1 >> 8) & 0xFF), x3 = (int8_t)((e1 >> 16) & 0xFF), y3 = (int8_t)(e1 >> 24);
                const bool b0 = sdl_bit(g, bx + x0, by + y0), b1 = sdl_bit(g, bx + x1, by + y1);
                const bool b2 = sdl_bit(g, bx + x2, by + y2), b3 = sdl_bit(g, bx + x3, by + y3);
                if (b0 | b1 | b2 | b3) {
                    s.px = ox + (b0 ? x0 : b1 ? x1 : b2 ? x2 : x3);
                    s.py = oy + (b0 ? y0 : b1 ? y1 : b2 ? y2 : y3);
                    s.cert = true;
                    s.stage = 1;
                    s.upDone = s.dnDone = true;  // the next iteration goes straight to the end-of-query logic
                    break;
                }
            }
        }
        if (!s.cert && sd_side_is_empty(g, s.sx, s.sy, s.sx + s.ax, s.sy + s.ay, s.dir)) {
            s.stage = 2;                     // hull edge, certified by the bounding box of the sites
            s.upDone = s.dnDone = true;
        } else if (!s.cert && s.dir < 0) {
            s.stage = 3;                     // clockwise queries have no sweep: the site goes to the general walk
            s.upDone = s.dnDone = true;
        }
    } else {
        s.m = 0;
        s.hx = s.hy = 0.f;
        s.hr2 = 1e9f;
        s.ux = s.uy = 0.f;
        s.r2 = 1e9f;
    }
}

SD_FN void sdl_begin(SdLocal& s, const SdGrid& g, int sx, int sy) {
    s.sx = sx;
    s.sy = sy;
    s.deg = 0;
    s.dir = 1;
    s.ax = s.ay = 0;
    // an 8-neighbour, if there is one, is a nearest site (distance 1 before sqrt 2) and needs no search
    const uint32_t c = sdl_row32(g, sy, sx + SDL_XLO), u = sdl_row32(g, sy + 1, sx + SDL_XLO), d = sdl_row32(g, sy - 1, sx + SDL_XLO);
    const int o = 0 - SDL_XLO;
    int nx = SDL_NONE, ny = 0;
    if ((d >> (o + 1)) & 1u) { nx = 1; ny = -1; }
    if ((d >> (o - 1)) & 1u) { nx = -1; ny = -1; }
    if ((u >> (o - 1)) & 1u) { nx = -1; ny = 1; }
    if ((u >> (o + 1)) & 1u) { nx = 1; ny = 1; }
    if ((d >> o) & 1u) { nx = 0; ny = -1; }
    if ((u >> o) & 1u) { nx = 0; ny = 1; }
    if ((c >> (o - 1)) & 1u) { nx = -1; ny = 0; }
    if ((c >> (o + 1)) & 1u) { nx = 1; ny = 0; }
    // A site only emits the triangles whose other two vertices FOLLOW it in raster order, i.e. lie at angles [0, pi)
    // counted counter-clockwise from +x.  If the pixel to the right is a site it is the first such neighbour, and the
    // walk can stop as soon as it reaches a neighbour that precedes s: about half of the star is never computed.
    s.half = ((c >> (o + 1)) & 1u) != 0;
    if (nx != SDL_NONE) {
        s.n0x = s.ax = nx;
        s.n0y = s.ay = ny;
        sdl_start_query(s, g, SDL_MODE_APEX);
    } else {
        sdl_start_query(s, g, SDL_MODE_NEAREST);
    }
}

// exact in float32: all operands are integers below 2^24 (coordinates relative to s, |.| <= 31 between window points)
SD_FN bool sdl_inside(int ax, int ay, int px, int py, int cx, int cy) {
    // is c inside circle(s = origin, a, p), with (s, a, p) counter-clockwise ?  (perturbed, never a tie)
    const float adx = (float)(0 - cx), ady = (float)(0 - cy), bdx = (float)(ax - cx), bdy = (float)(ay - cy);
    const float cdx = (float)(px - cx), cdy = (float)(py - cy);
    const float ad = adx * adx + ady * ady, bd = bdx * bdx + bdy * bdy, cd = cdx * cdx + cdy * cdy;
    const float det = adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
    if (det != 0.f) return det > 0.f;
    // co-circular: the raster-first of s(0,0), a, p, c decides (see sd_inside)
    const bool s_first = sd_before(0, 0, ax, ay) && sd_before(0, 0, px, py) && sd_before(0, 0, cx, cy);
    const bool a_first = sd_before(ax, ay, 0, 0) && sd_before(ax, ay, px, py) && sd_before(ax, ay, cx, cy);
    const bool p_first = sd_before(px, py, 0, 0) && sd_before(px, py, ax, ay) && sd_before(px, py, cx, cy);
    if (s_first) return sd_orient(cx, cy, ax, ay, px, py) > 0;
    if (a_first) return sd_orient(0, 0, cx, cy, px, py) > 0;
    if (p_first) return sd_orient(0, 0, ax, ay, cx, cy) > 0;
    return false;
}

SD_FN void sdl_set_apex(SdLocal& s, int cx, int cy) {
    s.px = cx;
    s.py = cy;
    const float ax = (float)s.ax, ay = (float)s.ay, x = (float)cx, y = (float)cy;
    const float d = 2.f * (ax * y - ay * x);  // > 0: c is strictly left of s -> a
    const float a2 = ax * ax + ay * ay, c2 = x * x + y * y;
    const float inv_d = SDL_RCP(d);
    s.ux = (y * a2 - ay * c2) * inv_d;
    s.uy = (ax * c2 - x * a2) * inv_d;
    s.r2 = s.ux * s.ux + s.uy * s.uy;
}

// One iteration of the lane's state machine: (advance to the next window row if the current one is exhausted) and
// (test one candidate if there is one).  `emit(ax, ay, bx, by, cx, cy)` receives owned triangles (absolute
// coordinates, counter-clockwise).
template <class Emit>
SD_FN int sdl_iter(SdLocal& s, const SdGrid& g, Emit& emit) {
    if (s.bits == 0u) {
        // ---- advance to the next window row (zig-zag away from row m), or finish the query.
        //      Rows and columns are cut to the "mask circle": the candidate's circle once there is a candidate,
        //      the small search disc around the edge before that (stage 0), nothing in stage 1.
        const bool have = s.px != SDL_NONE;
        const float mx = have ? s.ux : s.hx, my = have ? s.uy : s.hy;
        const float mr2 = have ? s.r2 : (s.stage == 0 ? s.hr2 : 1e9f);
        const float rad = SDL_SQRT(mr2) * 1.000001f + 0.75f;
        int r = 0;
        bool found = false;
#pragma unroll
        for (int attempt = 0; attempt < 2 && !found && !(s.upDone && s.dnDone); attempt++) {
            s.k++;
            const int off = (s.k + 1) >> 1;
            const bool up = (s.k & 1) != 0;
            r = s.m + (up ? off : -off);
            if (up) {
                if (r > SDL_HALF || (float)r > my + rad) s.upDone = true;
                found = !s.upDone;
            } else {
                if (r < -SDL_HALF || (float)r < my - rad) s.dnDone = true;
                found = !s.dnDone;
            }
            if (s.upDone && s.dnDone) break;
        }
        if (found) {
            const float dy = (float)r - my;
            const float h2 = mr2 - dy * dy;
            // float32 round-off of h2 is a few ulp of r^2: widen by it so that the mask stays a superset
            const float half = SDL_SQRT((h2 > 0.f ? h2 : 0.f) + 4e-6f * mr2) * 1.000001f + 0.75f;
            int xl = (int)floorf(fmaxf(mx - half, -64.f)), xr = (int)ceilf(fminf(mx + half, 64.f));
            if (s.mode == SDL_MODE_APEX) {  // strictly left of s -> a:  ay * x < ax * y
                const float t = (float)(s.ax * r);
                if (s.ay > 0) {
                    const int b = (int)ceilf(t * s.inv_ay + 0.01f);  // x <= ceil(t/ay) is a superset of x < t/ay
                    if (b < xr) xr = b;
                } else if (s.ay < 0) {
                    const int b = (int)floorf(t * s.inv_ay - 0.01f);
                    if (b > xl) xl = b;
                } else if (t <= 0.f) {
                    xr = xl - 1;
                }
            }
            s.row = r;
            // the window row straight from the LDS bitmap (the site itself is not a candidate)
            s.bits = sdl_row32(g, s.sy + r, s.sx + SDL_XLO) & sdl_range_mask(xl, xr) & (r == 0 ? ~(1u << (0 - SDL_XLO)) : 0xFFFFFFFFu);
        } else if (s.upDone && s.dnDone) {
            // ---- sweep finished
            if (s.mode == SDL_MODE_NEAREST) {
                if (!have || s.r2 > (float)(SDL_HALF * SDL_HALF)) return SDL_HARD(0);  // a site outside the window could be nearer
                s.n0x = s.ax = s.px;
                s.n0y = s.ay = s.py;
                sdl_start_query(s, g, SDL_MODE_APEX);
                return SDL_CONTINUE;
            }
            if (s.stage == 3) return SDL_HARD(1);
            if (s.stage == 2) {
                // hull edge.  A half walk owns nothing beyond it; a full walk goes back to its first neighbour and
                // fans out clockwise until it meets the hull on the other side.
                if (s.half || s.dir < 0) return SDL_SITE_DONE;
                s.dir = -1;
                s.ax = s.n0x;
                s.ay = s.n0y;
                sdl_start_query(s, g, SDL_MODE_APEX);
                return SDL_CONTINUE;
            }
            if (s.stage == 0 && !s.cert) {
                // the stage-0 sweep saw every site of the search disc; is that all of the candidate's circle?
                bool enough = false;
                if (have) {
                    const float dx = s.ux - s.hx, dy = s.uy - s.hy;
                    const float gap = SDL_SQRT(s.hr2) - SDL_SQRT(s.r2) - 0.3f;
                    enough = gap > 0.f && dx * dx + dy * dy < gap * gap;
                }
                if (!enough) {
                    s.stage = 1;
                    sdl_restart_scan(s);
                    return SDL_CONTINUE;
                }
            }
            if (!have) return SDL_HARD(2);  // nothing in the window (an interior hull edge, or a far apex)
            if (!s.cert) {
                const float rr = SDL_SQRT(s.r2) * 1.000001f + 0.26f;
                if (s.ux - rr < (float)SDL_XLO || s.ux + rr > (float)SDL_XHI || s.uy - rr < (float)-SDL_HALF ||
                    s.uy + rr > (float)SDL_HALF)
                    return SDL_HARD(3);  // the circle leaves the window: not certified
            }
            // triangle (s, a, p) -- (s, p, a) when walking clockwise; s owns it iff it is the raster-first vertex
            if (sd_before(0, 0, s.ax, s.ay) && sd_before(0, 0, s.px, s.py)) {
                if (s.dir > 0) emit(s.sx, s.sy, s.sx + s.ax, s.sy + s.ay, s.sx + s.px, s.sy + s.py);
                else emit(s.sx, s.sy, s.sx + s.px, s.sy + s.py, s.sx + s.ax, s.sy + s.ay);
            }
            if (s.dir > 0 && s.px == s.n0x && s.py == s.n0y) return SDL_SITE_DONE;
            if (s.half && !sd_before(0, 0, s.px, s.py)) return SDL_SITE_DONE;  // the rest of the star belongs to other sites
            if (++s.deg > 64) return SDL_HARD(4);
            s.ax = s.px;
            s.ay = s.py;
            sdl_start_query(s, g, SDL_MODE_APEX);
            return SDL_CONTINUE;
        }
    }
    if (s.bits != 0u) {
        // ---- test one candidate
        const int b = sd_ctz(s.bits);
        s.bits &= s.bits - 1u;
        const int x = b + SDL_XLO, y = s.row;
        if (s.mode == SDL_MODE_NEAREST) {
            const float d2 = (float)(x * x + y * y);
            if (d2 < s.r2) {
                s.px = x;
                s.py = y;
                s.r2 = d2;
            }
        } else if (s.ax * y - s.ay * x > 0) {  // strictly left of s -> a
            if (s.px == SDL_NONE) {
                sdl_set_apex(s, x, y);
            } else {
                const float ex = (float)x - s.ux, ey = (float)y - s.uy;
                if (ex * ex + ey * ey <= s.r2 + 0.01f * (1.f + s.r2) && sdl_inside(s.ax, s.ay, s.px, s.py, x, y))
                    sdl_set_apex(s, x, y);
            }
        }
    }
    return SDL_CONTINUE;
}
