// star_local.h -- the LEAN Delaunay star walk: one site per lane, one apex query per loop iteration, every query answered
// by the pre-sorted candidate table of star_table.h (bitmap probes only) or by the bounding-box hull test.
//
// Why: with one site per lane and the textbook nested loops of star_delaunay.h (steps -> rows -> words -> bits) a
// wavefront pays, at every nesting level, for its slowest lane; measured lane utilisation of that form was 15 %.  A first
// remedy was a per-lane state machine that swept a 32 x 15 window with exact float32 predicates (18 us per render); the
// table made the sweeps unnecessary for short edges (5.7 us), and measurements showed that of the sites with a query the
// table cannot answer (1.6 %), 97 % end in the general walk anyway -- their empty circles are large.  So the lean walk
// keeps NO sweep at all: straight-line code, and a site it cannot finish (SDL_LEAN_HARD) is walked again from scratch by
// the wave-cooperative general algorithm (sd_star).  Triangles it had already emitted are then emitted twice;
// rasterising a triangle is idempotent.
//
// Exactness: the table order is the exact perturbed in-circle order (built on the host with sd_inside), complete up to
// its last entry, so the first occupied entry IS the apex; nothing is approximated here.
#pragma once
#include "star_delaunay.h"
#include "star_table.h"

#define SDL_XLO (-16)    // sdl_row32 window: columns sx-16 .. sx+15 (bit = dx + 16)
#define SDL_NONE 0x7FFF

// 32 bits of bitmap row y starting at column x0 (may be negative / beyond the image: zeros)
SD_FN uint32_t sdl_row32(const SdGrid& g, int y, int x0) {
    // branch-free, like sdl_bit: words beyond the image are read at index 0 and masked (each guarded load was an exec-mask
    // round trip, six of them per site in sdl_lean_begin)
    const bool yin = (unsigned)y < (unsigned)g.H;
    const int w0 = x0 >> 5, sh = x0 & 31;  // arithmetic shift: floor
    const bool in0 = yin & ((unsigned)w0 < (unsigned)g.wpr), in1 = yin & ((unsigned)(w0 + 1) < (unsigned)g.wpr);
    const int base = SD_MUL(yin ? y : 0, g.wpr);
    uint32_t lo = g.occ[base + (in0 ? w0 : 0)], hi = g.occ[base + (in1 ? w0 + 1 : 0)];
    lo = in0 ? lo : 0u;
    hi = in1 ? hi : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);   // (hi : lo) >> sh, sh = 0 .. 31
#else
    return sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
#endif
}

SD_FN bool sdl_bit(const SdGrid& g, int x, int y) {
    // branch-free: an out-of-image probe reads word 0 and is masked (a guarded load costs an exec-mask round trip per probe)
    const bool in = (unsigned)x < (unsigned)g.W && (unsigned)y < (unsigned)g.H;
    const int xi = in ? x : 0, yi = in ? y : 0;
    return in & (bool)((g.occ[SD_MUL(yi, g.wpr) + (xi >> 5)] >> (xi & 31)) & 1u);
}

// Sites that need no walk at all.  If (x+1, y), (x-1, y) and (x, y+1) are sites, then every triangle that s OWNS (s its
// raster-first vertex, i.e. its neighbours at angles [0, pi)) is a unit triangle: adjacent pixels are always Delaunay neighbours,
// so (1, 0), (0, 1) and (-1, 0) are neighbours of s; the circle through s, (1, 0), (0, 1) is the circle of a unit square -- no
// lattice point inside, only (1, 1) on it --, so between (1, 0) and (0, 1) the star of s holds either nothing (triangle
// s, (1, 0), (0, 1)) or the neighbour (1, 1) (whichever diagonal the perturbation picks), unit triangles either way; the same
// on the other side with (-1, 1); and the walk ends at (-1, 0), which precedes s.  A unit triangle (area 1/2) has no lattice
// point other than its vertices (Pick), so there is nothing to rasterise: such sites -- 60-70 % of them in a texture map -- are
// left out of the site list.  (Until round 3 the test also asked for (x-1, y+1) and (x+1, y+1): sufficient, not necessary; the
// weaker test leaves out a third more.)  Word-parallel: bit b of the result = site b of word `c` (row y; l, r its neighbour
// words, uc the same of row y + 1, zeros beyond the image) needs no walk.  tests/host checks the claim on the host.
SD_FN uint32_t sdl_surrounded_word(uint32_t c, uint32_t l, uint32_t r, uint32_t uc) {
    const uint32_t right = (c >> 1) | (r << 31), left = (c << 1) | (l >> 31);
    return c & right & left & uc;
}
// the sites of bitmap word i = y * wpr + w that DO need a walk
SD_FN uint32_t sdl_walk_word(const uint32_t* occ, int H, int wpr, int i) {
    const int y = i / wpr, w = i - y * wpr;
    const uint32_t c = occ[i];
    if (c == 0u || y + 1 >= H) return c;
    const uint32_t l = w > 0 ? occ[i - 1] : 0u, r = w + 1 < wpr ? occ[i + 1] : 0u;
    return c & ~sdl_surrounded_word(c, l, r, occ[i + wpr]);
}

enum { SDL_LEAN_CONTINUE = 0, SDL_LEAN_DONE = 1, SDL_LEAN_HARD = 2 };
#ifndef SD_HARD_REASON
#define SD_HARD_REASON(slot)
#endif

struct SdLean {
    int sx, sy;
    int n0x, n0y, ax, ay;  // first neighbour, current edge end (relative to s)
    int dir, deg;
    int round;             // table probe round of the current query (4 candidates per round)
    bool half;             // the walk started at the neighbour (+1, 0): stop at the first neighbour that precedes s in raster order
};

SD_FN int sdl_lean_begin(SdLean& s, const SdGrid& g, int sx, int sy) {
    s.sx = sx;
    s.sy = sy;
    s.deg = 0;
    s.dir = 1;
    s.round = 0;
    s.half = false;
    s.n0x = s.n0y = s.ax = s.ay = 0;
    if (g.tab == nullptr) return SDL_LEAN_HARD;
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
    s.half = ((c >> (o + 1)) & 1u) != 0;
    s.n0x = s.ax = nx;
    s.n0y = s.ay = ny;
    return nx == SDL_NONE ? SDL_LEAN_HARD : SDL_LEAN_CONTINUE;  // no 8-neighbour: the general walk finds the nearest site
}

template <class Emit>
SD_FN int sdl_lean_step(SdLean& s, const SdGrid& g, Emit& emit) {
    // One round of the table probe per call: four entries per table read, four independent bitmap probes.  A query whose
    // first four candidates are all empty (rare where the image is densely covered) simply takes another call, so lanes
    // never wait for each other inside a probe loop.  The right of s -> a is the left of a -> s: clockwise queries enter the
    // table with the edge reversed.
    const int ox = s.dir > 0 ? 0 : s.ax, oy = s.dir > 0 ? 0 : s.ay;
    const int vx = s.dir > 0 ? s.ax : -s.ax, vy = s.dir > 0 ? s.ay : -s.ay;
    int px = SDL_NONE, py = 0;
    if (vx >= -SDT_AMAX && vx <= SDT_AMAX && vy >= -SDT_AMAX && vy <= SDT_AMAX) {
        const uint32_t* row = (const uint32_t*)(g.tab + sdt_index(vx, vy) * (SDT_LEN * 2));
        const int bx = s.sx + ox, by = s.sy + oy;
        const int k = s.round;
        const uint32_t e0 = row[2 * k], e1 = row[2 * k + 1];
        const int x0 = (int8_t)(e0 & 0xFF), y0 = (int8_t)((e0 >> 8) & 0xFF), x1 = (int8_t)((e0 >> 16) & 0xFF), y1 = (int8_t)(e0 >> 24);
        const int x2 = (int8_t)(e1 & 0xFF), y2 = (int8_t)((e1 >> 8) & 0xFF), x3 = (int8_t)((e1 >> 16) & 0xFF), y3 = (int8_t)(e1 >> 24);
        const bool b0 = sdl_bit(g, bx + x0, by + y0), b1 = sdl_bit(g, bx + x1, by + y1);
        const bool b2 = sdl_bit(g, bx + x2, by + y2), b3 = sdl_bit(g, bx + x3, by + y3);
        if (b0 | b1 | b2 | b3) {
            px = ox + (b0 ? x0 : b1 ? x1 : b2 ? x2 : x3);
            py = oy + (b0 ? y0 : b1 ? y1 : b2 ? y2 : y3);
        } else if (k + 1 < SDT_LEAN_LEN / 4) {
            s.round = k + 1;  // same query, next four candidates
            return SDL_LEAN_CONTINUE;
        }
    }
    s.round = 0;
    if (px == SDL_NONE) {
        if (!sd_side_is_empty(g, s.sx, s.sy, s.sx + s.ax, s.sy + s.ay, s.dir)) {
            SD_HARD_REASON((vx >= -SDT_AMAX && vx <= SDT_AMAX && vy >= -SDT_AMAX && vy <= SDT_AMAX) ? 7 : 6);
            return SDL_LEAN_HARD;
        }
        // Hull edge.  A half walk owns nothing beyond it; a full walk goes back to its first neighbour and fans out
        // clockwise until it meets the hull on the other side.
        if (s.half || s.dir < 0) return SDL_LEAN_DONE;
        s.dir = -1;
        s.ax = s.n0x;
        s.ay = s.n0y;
        return SDL_LEAN_CONTINUE;
    }
    if (sd_before(0, 0, s.ax, s.ay) && sd_before(0, 0, px, py)) {
        if (s.dir > 0) emit(s.sx, s.sy, s.sx + s.ax, s.sy + s.ay, s.sx + px, s.sy + py);
        else emit(s.sx, s.sy, s.sx + px, s.sy + py, s.sx + s.ax, s.sy + s.ay);
    }
    if (s.dir > 0 && px == s.n0x && py == s.n0y) return SDL_LEAN_DONE;
    if (s.half && !sd_before(0, 0, px, py)) return SDL_LEAN_DONE;
    s.ax = px;
    s.ay = py;
    if (++s.deg > 64) return SDL_LEAN_HARD;
    return SDL_LEAN_CONTINUE;
}
