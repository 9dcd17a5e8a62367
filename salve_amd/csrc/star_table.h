// star_table.h -- table-driven apex queries for short Delaunay edges.
//
// For a directed Delaunay edge s -> a, the apex of the triangle on its left is the occupied lattice point c (strictly
// left of the edge) whose circle through s, a, c comes first when the circle is grown into the left half-plane --
// ties between co-circular points broken by the symbolic perturbation of star_delaunay.h.  That order depends only on
// the edge VECTOR a - s: translating all points leaves both the in-circle determinant and the raster order of any two
// points unchanged.  So for every short edge vector the candidates can be sorted ONCE, on the host, with the exact
// predicates, and an apex query becomes "probe the bitmap at these offsets, in this order, until one is occupied" --
// no geometry, no floating point, a handful of instructions per probe.  On a 65 %-occupied region the first or second
// probe hits.  If the first SDT_LEN entries are all empty the caller falls back to the sweep of star_local.h.
#pragma once
#include <stdint.h>
#include "star_delaunay.h"

#include <algorithm>
#include <vector>
// Host-side construction with the exact perturbed predicate.  Returns false if the construction could not be certified
// complete (a kept candidate too close to the search radius).
static inline bool sdt_build(SdTable* t) {
    for (int ay = -SDT_AMAX; ay <= SDT_AMAX; ay++) {
        for (int ax = -SDT_AMAX; ax <= SDT_AMAX; ax++) {
            int8_t(*dst)[2] = t->off[sdt_index(ax, ay)];
            for (int k = 0; k < SDT_LEN; k++) dst[k][0] = dst[k][1] = 0;
            if (ax == 0 && ay == 0) continue;
            struct P { int x, y; };
            std::vector<P> cand;
            for (int y = -SDT_REACH; y <= SDT_REACH; y++)
                for (int x = -SDT_REACH; x <= SDT_REACH; x++)
                    if (sd_orient(0, 0, ax, ay, x, y) > 0) cand.push_back({x, y});
            // c1 comes first iff c1 lies inside circle(s, a, c2): strict total order under the perturbation
            std::sort(cand.begin(), cand.end(), [&](const P& c1, const P& c2) {
                if (c1.x == c2.x && c1.y == c2.y) return false;
                return sd_inside(0, 0, ax, ay, c2.x, c2.y, c1.x, c1.y);
            });
            // completeness: every lattice point inside the circle of the last kept candidate must have been enumerated
            const SdCircle c = sd_circle(0, 0, ax, ay, cand[SDT_LEN - 1].x, cand[SDT_LEN - 1].y);
            const double r = c.rpad + 1.0;
            if (c.ox - r < -SDT_REACH || c.ox + r > SDT_REACH || c.oy - r < -SDT_REACH || c.oy + r > SDT_REACH) return false;
            for (int k = 0; k < SDT_LEN; k++) {
                if (cand[k].x < -127 || cand[k].x > 127 || cand[k].y < -127 || cand[k].y > 127) return false;
                dst[k][0] = (int8_t)cand[k].x;
                dst[k][1] = (int8_t)cand[k].y;
            }
        }
    }
    return true;
}
