// Development (CPU only): counters of the general star walk's apex queries on real site sets, from the SAME headers the densify
// kernel compiles (star_delaunay.h / star_local.h / star_table.h, SD_COUNT hooks), with the kernel's E2 schedule simulated: NW
// wavefronts take runs of RUN hard-list entries and advance site by site in lock step; what a wavefront enters in the triangle
// cache becomes visible to the OTHERS one site-step later.  Answers "would X reduce the sweeps?" in seconds, before a GPU run.
//   python tools/probe/host/walk_sites.py            -> /tmp/walk_sites/sites_<scene>_<j>.bin   (oracle renders of synthetic scenes)
//   g++ -O2 [-DORDER=1 -DBLK=16] [-DNW=8 -DRUN=8] -o /tmp/walk_counters tools/probe/host/walk_counters.cpp
//   /tmp/walk_counters /tmp/walk_sites/sites_*.bin
// ORDER: 0 = the hard list in the order the lean walks gave up (raster order), 1 = sorted by BLK x BLK pixel block, 2 = Morton order.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static long long g_apex, g_apex_table, g_apex_cached, g_apex_far, g_apex_slow, g_rows, g_bits, g_exact;
#define SD_COUNT(c) (g_##c++)
#include "../../../salve_amd/csrc/star_delaunay.h"
#include "../../../salve_amd/csrc/star_table.h"
#include "../../../salve_amd/csrc/star_local.h"
#ifndef ORDER
#define ORDER 0
#endif
#ifndef BLK
#define BLK 16
#endif
#ifndef NW
#define NW 8
#endif
#ifndef RUN
#define RUN 8
#endif
struct HardEnt { int x, y; bool fresh; int ax, ay, dir; bool half; int n0x, n0y; };
static unsigned morton(unsigned x, unsigned y) { unsigned m = 0; for (int b = 0; b < 10; b++) m |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1); return m; }
struct Collect { std::vector<int>* out; void operator()(int ax, int ay, int bx, int by, int cx, int cy) { int v[6] = {ax, ay, bx, by, cx, cy}; out->insert(out->end(), v, v + 6); } };

int main(int argc, char** argv) {
    static SdTable t;
    if (!sdt_build(&t)) { fprintf(stderr, "candidate table not certified\n"); return 1; }
    static unsigned long long cache[SD_CACHE_SIZE], pend[NW][SD_CACHE_SIZE], view[SD_CACHE_SIZE];
    for (int f = 1; f < argc; f++) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) { perror(argv[f]); return 1; }
        std::vector<int> xy; int v[2];
        while (fread(v, 4, 2, fp) == 2) { xy.push_back(v[0]); xy.push_back(v[1]); }
        fclose(fp);
        const int n = (int)xy.size() / 2, H = 501, W = 501, wpr = (W + 31) / 32;
        std::vector<uint32_t> occ((size_t)H * wpr, 0);
        std::vector<int16_t> rmin(H, (int16_t)W), rmax(H, (int16_t)-1);
        int bx0 = W, bx1 = -1, by0 = H, by1 = -1;
        for (int i = 0; i < n; i++) {
            const int x = xy[2 * i], y = xy[2 * i + 1];
            occ[(size_t)y * wpr + (x >> 5)] |= 1u << (x & 31);
            if (x < rmin[y]) rmin[y] = (int16_t)x;
            if (x > rmax[y]) rmax[y] = (int16_t)x;
            bx0 = std::min(bx0, x); bx1 = std::max(bx1, x); by0 = std::min(by0, y); by1 = std::max(by1, y);
        }
        memset(cache, 0, sizeof(cache)); memset(pend, 0, sizeof(pend));
        SdGrid g = {H, W, wpr, occ.data(), rmin.data(), rmax.data(), 0, 1, &t.off[0][0][0], bx0, bx1, by0, by1, cache};
        g_apex = g_apex_table = g_apex_cached = g_apex_far = g_apex_slow = g_rows = g_bits = g_exact = 0;
        std::vector<int> out; Collect c = {&out}; std::vector<HardEnt> hl;
        for (int i = 0; i < n; i++) {   // E1: the lean walks
            SdLean ls; long long it = 0;
            int r = sdl_lean_begin(ls, g, xy[2 * i], xy[2 * i + 1]);
            while (r == SDL_LEAN_CONTINUE && it < 100000) { r = sdl_lean_step(ls, g, c); it++; }
            if (r != SDL_LEAN_DONE) { HardEnt h = {xy[2 * i], xy[2 * i + 1], ls.n0x == SDL_NONE, ls.ax, ls.ay, ls.dir, ls.half, ls.n0x, ls.n0y}; hl.push_back(h); }
        }
        if (ORDER == 1) std::stable_sort(hl.begin(), hl.end(), [](const HardEnt& a, const HardEnt& b) { return (a.y / BLK) * 64 + a.x / BLK < (b.y / BLK) * 64 + b.x / BLK; });
        if (ORDER == 2) std::stable_sort(hl.begin(), hl.end(), [](const HardEnt& a, const HardEnt& b) { return morton(a.x, a.y) < morton(b.x, b.y); });
        int next = 0, cur[NW], endr[NW];
        for (int w = 0; w < NW; w++) cur[w] = endr[w] = 0;
        for (;;) {   // E2: the general walks, NW wavefronts in lock step
            bool any = false;
            for (int w = 0; w < NW; w++) {
                if (cur[w] >= endr[w]) { if (next >= (int)hl.size()) continue; cur[w] = next; endr[w] = std::min<int>(next + RUN, (int)hl.size()); next = endr[w]; }
                any = true;
                const HardEnt& h = hl[cur[w]++];
                for (int k = 0; k < SD_CACHE_SIZE; k++) view[k] = pend[w][k] ? pend[w][k] : cache[k];   // the shared cache + this wavefront's own entries
                SdGrid gw = g; gw.cache = view;
                if (h.fresh) sd_star(gw, h.x, h.y, c); else sd_star_resume(gw, h.x, h.y, h.ax, h.ay, h.dir, h.half, h.n0x, h.n0y, c);
                for (int k = 0; k < SD_CACHE_SIZE; k++) if (view[k] != (pend[w][k] ? pend[w][k] : cache[k])) pend[w][k] = view[k];
            }
            if (!any) break;
            for (int w = 0; w < NW; w++) for (int k = 0; k < SD_CACHE_SIZE; k++) if (pend[w][k]) { cache[k] = pend[w][k]; pend[w][k] = 0; }
        }
        // (the triangle list is order dependent only in its ORDER: hash the sorted triangles)
        std::vector<std::vector<int>> tris; for (size_t k = 0; k + 5 < out.size(); k += 6) tris.push_back(std::vector<int>(out.begin() + k, out.begin() + k + 6));
        std::sort(tris.begin(), tris.end());
        unsigned long long hsum = 0; for (auto& tr : tris) for (int q : tr) hsum = hsum * 1000003ull + (unsigned)q;
        printf("%s: sites %d triangles %zu (hash %llx) hard sites %zu | apex queries %lld = table %lld + cache %lld + sweeping %lld | window growths %lld circle sweeps %lld | lane-rows %lld candidates %lld exact predicates %lld\n",
               argv[f], n, tris.size(), hsum, hl.size(), g_apex, g_apex_table, g_apex_cached, g_apex - g_apex_table - g_apex_cached, g_apex_far, g_apex_slow, g_rows, g_bits, g_exact);
    }
    return 0;
}
