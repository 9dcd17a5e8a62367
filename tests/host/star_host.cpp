// Host build of salve_amd/csrc/star_delaunay.h (the SAME source the HIP kernel compiles) so that the
// star-wrapping algorithm can be unit-tested against the oracle without a GPU.  Test-only.
#include <vector>
#include <cstring>
#include "../../salve_amd/csrc/star_delaunay.h"
#include "../../salve_amd/csrc/star_table.h"

static const int8_t* host_table() {
    static SdTable t;
    static bool ok = sdt_build(&t);
    return ok ? &t.off[0][0][0] : nullptr;
}
extern "C" int star_host_table_ok() { return host_table() != nullptr; }
static int g_use_table = 1;
static int g_cache_on = 0;
static unsigned long long g_cache_mem[SD_CACHE_SIZE];
extern "C" void star_host_use_cache(int on) { g_cache_on = on; memset(g_cache_mem, 0, sizeof(g_cache_mem)); }
extern "C" void star_host_use_table(int on) { g_use_table = on; }

struct Collect {
    std::vector<int>* out;
    void operator()(int ax, int ay, int bx, int by, int cx, int cy) {
        int v[6] = {ax, ay, bx, by, cx, cy};
        out->insert(out->end(), v, v + 6);
    }
};

extern "C" int star_host_triangulate(const int* xs, const int* ys, int n, int H, int W, int* tri_xy, int cap,
                                     long long* total_steps) {
    const bool use_table = g_use_table != 0;
    if (use_table && !host_table()) return -3;
    if (g_cache_on) memset(g_cache_mem, 0, sizeof(g_cache_mem));  // entries are only valid for one point set
    int wpr = (W + 31) / 32;
    std::vector<uint32_t> occ((size_t)H * wpr, 0);
    std::vector<int16_t> rmin(H, (int16_t)W), rmax(H, (int16_t)-1);
    for (int i = 0; i < n; i++) {
        occ[(size_t)ys[i] * wpr + (xs[i] >> 5)] |= 1u << (xs[i] & 31);
        if (xs[i] < rmin[ys[i]]) rmin[ys[i]] = (int16_t)xs[i];
        if (xs[i] > rmax[ys[i]]) rmax[ys[i]] = (int16_t)xs[i];
    }
    int bx0 = W, bx1 = -1, by0 = H, by1 = -1;
    for (int i = 0; i < n; i++) {
        if (xs[i] < bx0) bx0 = xs[i];
        if (xs[i] > bx1) bx1 = xs[i];
        if (ys[i] < by0) by0 = ys[i];
        if (ys[i] > by1) by1 = ys[i];
    }
    SdGrid g = {H, W, wpr, occ.data(), rmin.data(), rmax.data(), 0, 1, use_table ? host_table() : nullptr, bx0, bx1, by0, by1, g_cache_on ? g_cache_mem : nullptr};
    std::vector<int> out;
    Collect c = {&out};
    long long steps = 0;
    for (int i = 0; i < n; i++) {
        int s = sd_star(g, xs[i], ys[i], c);
        if (s < 0) return -1;
        steps += s;
    }
    if (total_steps) *total_steps = steps;
    int nt = (int)(out.size() / 6);
    if (nt > cap) return -2;
    memcpy(tri_xy, out.data(), out.size() * sizeof(int));
    return nt;
}

#include "../../salve_amd/csrc/star_local.h"

// Lean walk (star_local.h) for every site, general algorithm for the sites it hands over -- the kernel's two phases.
// stats: [0] lean iterations, [1] hard sites, [2] max iterations of one site.
extern "C" int star_host_triangulate_local(const int* xs, const int* ys, int n, int H, int W, int* tri_xy, int cap,
                                           long long* stats) {
    const bool use_table = g_use_table != 0;
    if (use_table && !host_table()) return -3;
    if (g_cache_on) memset(g_cache_mem, 0, sizeof(g_cache_mem));  // entries are only valid for one point set
    int wpr = (W + 31) / 32;
    std::vector<uint32_t> occ((size_t)H * wpr, 0);
    std::vector<int16_t> rmin(H, (int16_t)W), rmax(H, (int16_t)-1);
    int bx0 = W, bx1 = -1, by0 = H, by1 = -1;
    for (int i = 0; i < n; i++) {
        occ[(size_t)ys[i] * wpr + (xs[i] >> 5)] |= 1u << (xs[i] & 31);
        if (xs[i] < rmin[ys[i]]) rmin[ys[i]] = (int16_t)xs[i];
        if (xs[i] > rmax[ys[i]]) rmax[ys[i]] = (int16_t)xs[i];
        if (xs[i] < bx0) bx0 = xs[i];
        if (xs[i] > bx1) bx1 = xs[i];
        if (ys[i] < by0) by0 = ys[i];
        if (ys[i] > by1) by1 = ys[i];
    }
    SdGrid g = {H, W, wpr, occ.data(), rmin.data(), rmax.data(), 0, 1, use_table ? host_table() : nullptr, bx0, bx1, by0, by1, g_cache_on ? g_cache_mem : nullptr};
    std::vector<int> out;
    Collect c = {&out};
    long long iters = 0, hard = 0, maxit = 0;
    for (int i = 0; i < n; i++) {
        std::vector<int> mine;
        Collect cm = {&mine};
        SdLean ls;
        long long it = 0;
        int r = sdl_lean_begin(ls, g, xs[i], ys[i]);
        while (r == SDL_LEAN_CONTINUE && it < 100000) { r = sdl_lean_step(ls, g, cm); it++; }
        iters += it;
        if (it > maxit) maxit = it;
        if (r == SDL_LEAN_DONE) {
            out.insert(out.end(), mine.begin(), mine.end());
        } else {
            hard++;   // as in the kernel: what the lean walk emitted stays, the general walk takes over at the edge it gave up on
            out.insert(out.end(), mine.begin(), mine.end());
            const bool fresh = ls.n0x == SDL_NONE || g.tab == nullptr;   // gave up in sdl_lean_begin: nothing to take over
            if (fresh) {
                if (sd_star(g, xs[i], ys[i], c) < 0) return -1;
            } else if (sd_star_resume(g, xs[i], ys[i], ls.ax, ls.ay, ls.dir, ls.half, ls.n0x, ls.n0y, c) < 0) {
                return -1;
            }
        }
    }
    if (stats) { stats[0] = iters; stats[1] = hard; stats[2] = maxit; }
    int nt = (int)(out.size() / 6);
    if (nt > cap) return -2;
    memcpy(tri_xy, out.data(), out.size() * sizeof(int));
    return nt;
}

// The sites the kernel leaves out of its list (sdl_walk_word): walk each of them with the general algorithm and count the
// emitted triangles that are NOT unit triangles (twice the area != 1) -- the claim is that there are none.
// stats: [0] sites left out, [1] triangles they own.
extern "C" int star_host_check_left_out(const int* xs, const int* ys, int n, int H, int W, long long* stats) {
    if (!host_table()) return -3;
    int wpr = (W + 31) / 32;
    std::vector<uint32_t> occ((size_t)H * wpr, 0);
    std::vector<int16_t> rmin(H, (int16_t)W), rmax(H, (int16_t)-1);
    int bx0 = W, bx1 = -1, by0 = H, by1 = -1;
    for (int i = 0; i < n; i++) {
        occ[(size_t)ys[i] * wpr + (xs[i] >> 5)] |= 1u << (xs[i] & 31);
        if (xs[i] < rmin[ys[i]]) rmin[ys[i]] = (int16_t)xs[i];
        if (xs[i] > rmax[ys[i]]) rmax[ys[i]] = (int16_t)xs[i];
        if (xs[i] < bx0) bx0 = xs[i];
        if (xs[i] > bx1) bx1 = xs[i];
        if (ys[i] < by0) by0 = ys[i];
        if (ys[i] > by1) by1 = ys[i];
    }
    SdGrid g = {H, W, wpr, occ.data(), rmin.data(), rmax.data(), 0, 1, host_table(), bx0, bx1, by0, by1, nullptr};
    long long left_out = 0, owned = 0;
    int bad = 0;
    for (int i = 0; i < H * wpr; i++) {
        uint32_t skip = occ[i] & ~sdl_walk_word(occ.data(), H, wpr, i);
        const int y = i / wpr, xb = (i % wpr) << 5;
        for (int b = 0; b < 32; b++) {
            if (!((skip >> b) & 1u)) continue;
            left_out++;
            std::vector<int> mine;
            Collect cm = {&mine};
            if (sd_star(g, xb + b, y, cm) < 0) return -1;
            for (size_t t = 0; t + 5 < mine.size(); t += 6) {
                owned++;
                const long long a2 = (long long)(mine[t + 2] - mine[t]) * (mine[t + 5] - mine[t + 1]) - (long long)(mine[t + 3] - mine[t + 1]) * (mine[t + 4] - mine[t]);
                if (a2 != 1 && a2 != -1) bad++;
            }
        }
    }
    if (stats) { stats[0] = left_out; stats[1] = owned; }
    return bad;
}
