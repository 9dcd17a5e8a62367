/*
 * oracle/csrc/delaunay_exact.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * CPU restatement of the sparse->dense densification step of the reference
 *   salve/utils/interpolation_utils.py:21-54  interp_dense_grid_from_sparse
 * which calls scipy.interpolate.griddata(method="linear") (interpolation_utils.py:46-48),
 * i.e. a Qhull Delaunay triangulation of the sparse BEV pixels followed by
 * barycentric interpolation at every integer grid point.  scipy / Qhull are a
 * third-party dependency of the reference (version unpinned by the reference,
 * scipy 1.15.3 in this image); the published algorithm is restated here:
 *
 *   1. Delaunay triangulation of the sites.  All sites are integer lattice points, so the
 *      orientation and in-circle determinants are evaluated EXACTLY in int64.
 *      Co-circular quadruples (ubiquitous on a lattice) make the Delaunay triangulation
 *      non-unique; Qhull's choice depends on input order and float round-off.  This
 *      restatement makes the triangulation unique by symbolic perturbation of the lifted
 *      height z_i = x_i^2 + y_i^2 + eps_i, eps_i > 0 infinitesimal with eps_i >> eps_j
 *      whenever site i precedes site j in raster order (y, then x).  Every correct
 *      Delaunay algorithm then produces the same triangles, which is what lets the HIP
 *      kernel (a completely different algorithm) be checked bit for bit.
 *   2. Linear interpolation in exact rational arithmetic: value = sum(w_k c_k) / sum(w_k)
 *      with integer barycentric weights; the uint8 result is the floor of that rational,
 *      which is what the reference's float64 -> uint8 truncating assignment
 *      (interpolation_utils.py:53) computes up to float round-off.
 *
 * Algorithm (independent of the GPU's star-wrapping algorithm): incremental insertion in
 * raster order ("sweep hull"): every new site is lexicographically largest, hence outside
 * the current hull; it is joined to all hull edges it sees and the new edges are legalised
 * by Lawson flips with the perturbed in-circle predicate.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int64_t i64;
typedef int32_t i32;

static inline i64 orient2d(i32 ax, i32 ay, i32 bx, i32 by, i32 cx, i32 cy) {
    return (i64)(bx - ax) * (i64)(cy - ay) - (i64)(by - ay) * (i64)(cx - ax);
}

/* > 0 iff d is strictly inside the circle through a, b, c (a, b, c counter-clockwise). */
static inline i64 incircle_det(i32 ax, i32 ay, i32 bx, i32 by, i32 cx, i32 cy, i32 dx, i32 dy) {
    i64 adx = ax - dx, ady = ay - dy;
    i64 bdx = bx - dx, bdy = by - dy;
    i64 cdx = cx - dx, cdy = cy - dy;
    i64 ad = adx * adx + ady * ady;
    i64 bd = bdx * bdx + bdy * bdy;
    i64 cd = cdx * cdx + cdy * cdy;
    return adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx);
}

typedef struct {
    const i32 *x, *y; /* sites, sorted by (y, x); the array index IS the perturbation rank */
} sites_t;

/* Perturbed in-circle: +1 inside, -1 outside, never 0 for four distinct sites with a,b,c ccw. */
static int incircle_sos(const sites_t *s, i32 a, i32 b, i32 c, i32 d) {
    i64 det = incircle_det(s->x[a], s->y[a], s->x[b], s->y[b], s->x[c], s->y[c], s->x[d], s->y[d]);
    if (det > 0) return 1;
    if (det < 0) return -1;
    /* co-circular: the site with the smallest rank carries the dominant perturbation */
    i32 m = a;
    if (b < m) m = b;
    if (c < m) m = c;
    if (d < m) m = d;
    i64 o;
    if (m == d) return -1;
    if (m == a) o = orient2d(s->x[d], s->y[d], s->x[b], s->y[b], s->x[c], s->y[c]);
    else if (m == b) o = orient2d(s->x[a], s->y[a], s->x[d], s->y[d], s->x[c], s->y[c]);
    else o = orient2d(s->x[a], s->y[a], s->x[b], s->y[b], s->x[d], s->y[d]);
    return o > 0 ? 1 : -1;
}

typedef struct {
    i32 *tv;  /* tv[3t+i]: start vertex of half-edge 3t+i (triangle t = tv[3t], tv[3t+1], tv[3t+2], ccw) */
    i32 *opp; /* twin half-edge or -1 on the hull */
    i32 nt;
    i32 *hull_next, *hull_prev; /* circular ccw list over hull vertices */
    i32 *hull_he;               /* hull_he[u] = half-edge u -> hull_next[u] (its opp is -1) */
    i32 *stack;
    i32 stack_cap;
} mesh_t;

static inline i32 nxt(i32 e) { return (e % 3 == 2) ? e - 2 : e + 1; }
static inline i32 prv(i32 e) { return (e % 3 == 0) ? e + 2 : e - 1; }

static inline void set_opp(mesh_t *m, i32 e, i32 o) {
    m->opp[e] = o;
    if (o >= 0) m->opp[o] = e;
    else m->hull_he[m->tv[e]] = e;
}

static i32 add_tri(mesh_t *m, i32 a, i32 b, i32 c) {
    i32 t = m->nt++;
    m->tv[3 * t] = a;
    m->tv[3 * t + 1] = b;
    m->tv[3 * t + 2] = c;
    m->opp[3 * t] = m->opp[3 * t + 1] = m->opp[3 * t + 2] = -1;
    return t;
}

/* Lawson legalisation starting from half-edge e0 (explicit stack). */
static void legalize(mesh_t *m, const sites_t *s, i32 e0) {
    i32 sp = 0;
    m->stack[sp++] = e0;
    while (sp > 0) {
        i32 a = m->stack[--sp];
        i32 b = m->opp[a];
        if (b < 0) continue;
        /* triangle ta = (P, Q, S) with a = P->Q ; tb = (Q, P, U) with b = Q->P */
        i32 a1 = nxt(a), a2 = prv(a);
        i32 b1 = nxt(b), b2 = prv(b);
        i32 P = m->tv[a], Q = m->tv[a1], S = m->tv[a2], U = m->tv[b2];
        if (incircle_sos(s, P, Q, S, U) <= 0) continue;
        i32 oa1 = m->opp[a1], oa2 = m->opp[a2], ob1 = m->opp[b1], ob2 = m->opp[b2];
        i32 ta = a / 3, tb = b / 3;
        /* new ta = (S, P, U): S->P (old a2), P->U (old b1), U->S (diagonal)
           new tb = (U, Q, S): U->Q (old b2), Q->S (old a1), S->U (diagonal) */
        m->tv[3 * ta] = S; m->tv[3 * ta + 1] = P; m->tv[3 * ta + 2] = U;
        m->tv[3 * tb] = U; m->tv[3 * tb + 1] = Q; m->tv[3 * tb + 2] = S;
        set_opp(m, 3 * ta, oa2);
        set_opp(m, 3 * ta + 1, ob1);
        set_opp(m, 3 * tb, ob2);
        set_opp(m, 3 * tb + 1, oa1);
        m->opp[3 * ta + 2] = 3 * tb + 2;
        m->opp[3 * tb + 2] = 3 * ta + 2;
        if (sp + 4 > m->stack_cap) {
            m->stack_cap *= 2;
            m->stack = (i32 *)realloc(m->stack, sizeof(i32) * (size_t)m->stack_cap);
        }
        m->stack[sp++] = 3 * ta;
        m->stack[sp++] = 3 * ta + 1;
        m->stack[sp++] = 3 * tb;
        m->stack[sp++] = 3 * tb + 1;
    }
}

/*
 * Triangulate n sites sorted by (y, x), all distinct.
 * tri_out: capacity 3 * 2n int32; returns number of triangles, 0 if all sites are collinear
 * (or n < 3), -1 on allocation failure.
 */
int salve_oracle_delaunay(const i32 *x, const i32 *y, i32 n, i32 *tri_out) {
    if (n < 3) return 0;
    sites_t s = {x, y};
    /* first site that is not collinear with sites 0 and 1 */
    i32 k = 2;
    while (k < n && orient2d(x[0], y[0], x[1], y[1], x[k], y[k]) == 0) k++;
    if (k == n) return 0;
    mesh_t m;
    size_t cap = (size_t)2 * (size_t)n + 8;
    m.tv = (i32 *)malloc(sizeof(i32) * 3 * cap);
    m.opp = (i32 *)malloc(sizeof(i32) * 3 * cap);
    m.hull_next = (i32 *)malloc(sizeof(i32) * (size_t)n);
    m.hull_prev = (i32 *)malloc(sizeof(i32) * (size_t)n);
    m.hull_he = (i32 *)malloc(sizeof(i32) * (size_t)n);
    m.stack_cap = 1024;
    m.stack = (i32 *)malloc(sizeof(i32) * (size_t)m.stack_cap);
    m.nt = 0;
    if (!m.tv || !m.opp || !m.hull_next || !m.hull_prev || !m.hull_he || !m.stack) return -1;

    /* sites 0..k-1 are collinear and ordered along their line; fan them to site k */
    int ccw = orient2d(x[0], y[0], x[1], y[1], x[k], y[k]) > 0;
    for (i32 i = 0; i + 1 < k; i++) {
        i32 t = ccw ? add_tri(&m, i, i + 1, k) : add_tri(&m, i + 1, i, k);
        (void)t;
    }
    /* link fan neighbours: triangle i shares edge (i+1, k) with triangle i+1 */
    for (i32 i = 0; i + 2 < k; i++) {
        if (ccw) { /* t_i = (i, i+1, k): edge 1 = (i+1 -> k); t_{i+1} = (i+1, i+2, k): edge 2 = (k -> i+1) */
            m.opp[3 * i + 1] = 3 * (i + 1) + 2;
            m.opp[3 * (i + 1) + 2] = 3 * i + 1;
        } else { /* t_i = (i+1, i, k): edge 2 = (k -> i+1); t_{i+1} = (i+2, i+1, k): edge 1 = (i+1 -> k) */
            m.opp[3 * i + 2] = 3 * (i + 1) + 1;
            m.opp[3 * (i + 1) + 1] = 3 * i + 2;
        }
    }
    /* hull, counter-clockwise */
    if (ccw) { /* 0 -> 1 -> ... -> k-1 -> k -> 0 */
        for (i32 i = 0; i < k; i++) { m.hull_next[i] = i + 1; m.hull_prev[i + 1] = i; }
        m.hull_next[k] = 0; m.hull_prev[0] = k;
        for (i32 i = 0; i + 1 < k; i++) m.hull_he[i] = 3 * i;             /* i -> i+1 */
        m.hull_he[k - 1] = 3 * (k - 2) + 1;                               /* k-1 -> k */
        m.hull_he[k] = 3 * 0 + 2;                                         /* k -> 0 */
    } else { /* k-1 -> k-2 -> ... -> 0 -> k -> k-1 */
        for (i32 i = k - 1; i > 0; i--) { m.hull_next[i] = i - 1; m.hull_prev[i - 1] = i; }
        m.hull_next[0] = k; m.hull_prev[k] = 0;
        m.hull_next[k] = k - 1; m.hull_prev[k - 1] = k;
        for (i32 i = 0; i + 1 < k; i++) m.hull_he[i + 1] = 3 * i;         /* i+1 -> i */
        m.hull_he[0] = 3 * 0 + 1;                                         /* 0 -> k */
        m.hull_he[k] = 3 * (k - 2) + 2;                                   /* k -> k-1 */
    }
    /* the fan may already violate Delaunay (only across its interior edges) */
    for (i32 i = 0; i + 2 < k; i++) legalize(&m, &s, ccw ? 3 * i + 1 : 3 * i + 2);

    i32 last = k; /* most recently inserted site: always a hull vertex */
    for (i32 p = k + 1; p < n; p++) {
        /* p is outside the hull and `last` is the hull vertex just before it in raster order, so one of the
           two hull edges at `last` is visible from p; extend in both directions */
        i32 start = last; /* edge start -> hull_next[start] */
        #define VISIBLE(u) (orient2d(x[u], y[u], x[m.hull_next[u]], y[m.hull_next[u]], x[p], y[p]) < 0)
        if (!VISIBLE(start)) {
            start = m.hull_prev[last];
            if (!VISIBLE(start)) {
                /* cannot happen for a lexicographically largest point; fall back to a full scan */
                i32 u = m.hull_next[last];
                while (u != last && !VISIBLE(u)) u = m.hull_next[u];
                start = u;
            }
        }
        i32 first = start;
        while (VISIBLE(m.hull_prev[first])) first = m.hull_prev[first];
        i32 end = m.hull_next[start]; /* one past the last visible edge's start */
        while (VISIBLE(end)) end = m.hull_next[end];
        #undef VISIBLE
        /* visible edges: u -> next(u) for u = first .. prev(end) */
        i32 prev_tri = -1;
        i32 u = first;
        i32 first_tri = -1;
        while (u != end) {
            i32 v = m.hull_next[u];
            i32 inner = m.hull_he[u]; /* half-edge u -> v of the interior triangle */
            i32 t = add_tri(&m, v, u, p); /* edges: 0: v->u, 1: u->p, 2: p->v */
            m.opp[3 * t] = inner;
            m.opp[inner] = 3 * t;
            if (prev_tri >= 0) { /* previous triangle (u, u_prev, p): its edge 2 = p -> u ; ours edge 1 = u -> p */
                m.opp[3 * t + 1] = 3 * prev_tri + 2;
                m.opp[3 * prev_tri + 2] = 3 * t + 1;
            } else {
                first_tri = t;
            }
            prev_tri = t;
            u = v;
        }
        /* new hull: first -> p -> end */
        m.hull_next[first] = p; m.hull_prev[p] = first;
        m.hull_next[p] = end;   m.hull_prev[end] = p;
        m.hull_he[first] = 3 * first_tri + 1; /* first -> p */
        m.hull_he[p] = 3 * prev_tri + 2;      /* p -> end */
        /* legalise the old hull edges that are now interior */
        for (i32 t = first_tri; t <= prev_tri; t++) legalize(&m, &s, 3 * t);
        last = p;
    }
    memcpy(tri_out, m.tv, sizeof(i32) * 3 * (size_t)m.nt);
    int nt = m.nt;
    free(m.tv); free(m.opp); free(m.hull_next); free(m.hull_prev); free(m.hull_he); free(m.stack);
    return nt;
}

/*
 * Exact barycentric rasterisation of the triangulation.
 *   col:     [n,3] uint8 colours of the sites
 *   out_u8:  [H,W,3] floor of the exact rational interpolant (0 outside the hull)
 *   out_f64: [H,W,3] the rational rounded to double (NaN outside the hull), may be NULL
 *   cover:   [H,W] uint8, 1 inside/on the hull, may be NULL
 */
void salve_oracle_rasterize(const i32 *x, const i32 *y, const uint8_t *col, const i32 *tri, i32 nt,
                            i32 H, i32 W, uint8_t *out_u8, double *out_f64, uint8_t *cover) {
    size_t npx = (size_t)H * (size_t)W;
    memset(out_u8, 0, npx * 3);
    if (cover) memset(cover, 0, npx);
    if (out_f64) for (size_t i = 0; i < npx * 3; i++) out_f64[i] = NAN;
    for (i32 t = 0; t < nt; t++) {
        i32 a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        i32 ax = x[a], ay = y[a], bx = x[b], by = y[b], cx = x[c], cy = y[c];
        i64 area = orient2d(ax, ay, bx, by, cx, cy);
        if (area <= 0) continue;
        i32 x0 = ax < bx ? ax : bx; if (cx < x0) x0 = cx;
        i32 x1 = ax > bx ? ax : bx; if (cx > x1) x1 = cx;
        i32 y0 = ay < by ? ay : by; if (cy < y0) y0 = cy;
        i32 y1 = ay > by ? ay : by; if (cy > y1) y1 = cy;
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
        if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
        for (i32 qy = y0; qy <= y1; qy++) {
            for (i32 qx = x0; qx <= x1; qx++) {
                i64 wa = orient2d(bx, by, cx, cy, qx, qy);
                i64 wb = orient2d(cx, cy, ax, ay, qx, qy);
                i64 wc = orient2d(ax, ay, bx, by, qx, qy);
                if (wa < 0 || wb < 0 || wc < 0) continue;
                size_t px = (size_t)qy * (size_t)W + (size_t)qx;
                if (cover) cover[px] = 1;
                for (int ch = 0; ch < 3; ch++) {
                    i64 num = wa * col[3 * a + ch] + wb * col[3 * b + ch] + wc * col[3 * c + ch];
                    out_u8[3 * px + ch] = (uint8_t)(num / area);
                    if (out_f64) out_f64[3 * px + ch] = (double)num / (double)area;
                }
            }
        }
    }
}

/* 1 iff no site lies strictly inside the circumcircle of any triangle (brute force; test sizes only). */
int salve_oracle_check_delaunay(const i32 *x, const i32 *y, i32 n, const i32 *tri, i32 nt) {
    for (i32 t = 0; t < nt; t++) {
        i32 a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        if (orient2d(x[a], y[a], x[b], y[b], x[c], y[c]) <= 0) return 0;
        for (i32 d = 0; d < n; d++) {
            if (d == a || d == b || d == c) continue;
            if (incircle_det(x[a], y[a], x[b], y[b], x[c], y[c], x[d], y[d]) > 0) return 0;
        }
    }
    return 1;
}

/* y = fma(b, r01, a * r00): the order OpenBLAS' FMA dgemm micro-kernel evaluates an inner dimension of two,
 * which is what `xy @ R.T` does in the reference (bev_rendering_utils.py:445-451, sim2.py:157). */
void salve_oracle_rot2(const double *xy_in, i64 n, const double *R, const double *t, double *xy_out) {
    for (i64 i = 0; i < n; i++) {
        double a = xy_in[2 * i], b = xy_in[2 * i + 1];
        double ox = fma(b, R[1], a * R[0]);
        double oy = fma(b, R[3], a * R[2]);
        if (t) { ox = ox + t[0]; oy = oy + t[1]; }
        xy_out[2 * i] = ox;
        xy_out[2 * i + 1] = oy;
    }
}

/*
 * flags[t] = 1 iff some other site lies exactly ON the circumcircle of triangle t (the triangle is then
 * not "strongly" Delaunay and a different, equally valid Delaunay triangulation may not contain it).
 * Used only to classify pixels into parity tiers against scipy.  occ is an [H,W] 0/1 occupancy image.
 */
void salve_oracle_tri_degenerate(const i32 *x, const i32 *y, const i32 *tri, i32 nt, const uint8_t *occ,
                                 i32 H, i32 W, uint8_t *flags) {
    for (i32 t = 0; t < nt; t++) {
        i32 a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        double ax = x[a], ay = y[a], bx = x[b], by = y[b], cx = x[c], cy = y[c];
        double d = 2.0 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by));
        double ux = ((ax * ax + ay * ay) * (by - cy) + (bx * bx + by * by) * (cy - ay) + (cx * cx + cy * cy) * (ay - by)) / d;
        double uy = ((ax * ax + ay * ay) * (cx - bx) + (bx * bx + by * by) * (ax - cx) + (cx * cx + cy * cy) * (bx - ax)) / d;
        double r = sqrt((ux - ax) * (ux - ax) + (uy - ay) * (uy - ay));
        double fy0 = floor(uy - r - 1), fy1 = ceil(uy + r + 1);
        i32 y0 = fy0 < 0 ? 0 : (fy0 > H - 1 ? H : (i32)fy0);
        i32 y1 = fy1 > H - 1 ? H - 1 : (fy1 < 0 ? -1 : (i32)fy1);
        uint8_t f = 0;
        for (i32 qy = y0; qy <= y1 && !f; qy++) {
            double dy = qy - uy;
            double h2 = r * r - dy * dy;
            double h = h2 > 0 ? sqrt(h2) : 0;
            /* only lattice points near the two circle crossings of this row can be on the circle */
            for (int side = 0; side < 2 && !f; side++) {
                double xc = side ? ux + h : ux - h;
                double lo = floor(xc - 2), hi = ceil(xc + 2);
                if (h2 <= 0) { lo = floor(ux - 2 - sqrt(2 * r + 4)); hi = ceil(ux + 2 + sqrt(2 * r + 4)); }
                i32 x0 = lo < 0 ? 0 : (lo > W - 1 ? W : (i32)lo);
                i32 x1 = hi > W - 1 ? W - 1 : (hi < 0 ? -1 : (i32)hi);
                for (i32 qx = x0; qx <= x1; qx++) {
                    if (!occ[(size_t)qy * W + qx]) continue;
                    if ((qx == x[a] && qy == y[a]) || (qx == x[b] && qy == y[b]) || (qx == x[c] && qy == y[c])) continue;
                    if (incircle_det(x[a], y[a], x[b], y[b], x[c], y[c], qx, qy) == 0) { f = 1; break; }
                }
            }
        }
        flags[t] = f;
    }
}
