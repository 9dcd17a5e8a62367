"""The GPU's Delaunay star algorithms (salve_amd/csrc/star_delaunay.h, star_local.h), compiled for the HOST with g++
from the very same headers, against the independent CPU oracle (incremental insertion + flips)."""

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import bev_oracle as bo
from salve_amd import synthetic

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = tmp_path_factory.mktemp("star") / "star_host.so"
    # STAR_HOST_CXXFLAGS: extra flags, e.g. "-fsanitize=undefined -fno-sanitize-recover=all" (sanitizers run on the CPU build only)
    extra = os.environ.get("STAR_HOST_CXXFLAGS", "").split()
    subprocess.run(["g++", "-O2", "-shared", "-fPIC"] + extra + ["-o", str(so), str(ROOT / "tests" / "host" / "star_host.cpp")], check=True)
    return ctypes.CDLL(str(so))


def run(fn, pts, H, W):
    xs = np.ascontiguousarray(pts[:, 0], dtype=np.int32)
    ys = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
    cap = 2 * len(xs) + 16
    out = np.zeros((cap, 6), dtype=np.int32)
    stats = np.zeros(4, dtype=np.int64)
    nt = fn(xs.ctypes.data_as(ctypes.c_void_p), ys.ctypes.data_as(ctypes.c_void_p), len(xs), H, W, out.ctypes.data_as(ctypes.c_void_p), cap,
            stats.ctypes.data_as(ctypes.c_void_p))
    assert nt >= 0
    v = out[:nt].reshape(-1, 3, 2)
    return sorted(map(tuple, np.sort(v[:, :, 1] * 4096 + v[:, :, 0], 1).tolist())), stats


def oracle_tris(pts):
    order, tri = bo.delaunay_exact(pts[:, 0], pts[:, 1])
    sp = pts[order]
    return sorted(map(tuple, np.sort(sp[tri][:, :, 1] * 4096 + sp[tri][:, :, 0], 1).tolist()))


def test_candidate_table_is_certified(lib):
    """star_table.h: the pre-sorted apex candidates of short edges were enumerated completely."""
    assert lib.star_host_table_ok() == 1


@pytest.mark.parametrize("table", [1, 0])
@pytest.mark.parametrize("which", ["star_host_triangulate", "star_host_triangulate_local"])
def test_random_lattice_sets(lib, which, table):
    lib.star_host_use_table(table)
    rng = np.random.default_rng(5)
    done = 0
    for _ in range(250):
        G = int(rng.integers(3, 60))
        pts = np.unique(rng.integers(0, G, size=(int(rng.integers(3, 500)), 2)), axis=0)
        if bo._is_degenerate(pts):
            continue
        ref = oracle_tris(pts)
        if not ref:
            continue
        got, _ = run(getattr(lib, which), pts, G, G)
        assert got == ref
        done += 1
    assert done > 150


def test_structured_degenerate_sets(lib):
    """Full lattices, lines with one off-point, rings: maximal co-circularity and collinear hulls."""
    yy, xx = np.mgrid[0:12, 0:17]
    full = np.stack([xx.ravel(), yy.ravel()], 1)
    line = np.array([[i, 3] for i in range(20)] + [[7, 9]])
    th = np.linspace(0, 2 * np.pi, 80, endpoint=False)
    ring = np.unique(np.round(np.stack([30 + 25 * np.cos(th), 30 + 25 * np.sin(th)], 1)).astype(int), axis=0)
    for pts, G in ((full, 17), (line, 20), (ring, 61), (full[::2], 17)):
        ref = oracle_tris(pts)
        for table in (1, 0):
            lib.star_host_use_table(table)
            for which in ("star_host_triangulate", "star_host_triangulate_local"):
                assert run(getattr(lib, which), pts, G, G)[0] == ref


def test_triangle_cache(lib):
    """The shared cache of resolved triangles (sd_cache_*) changes which queries sweep, never the result."""
    rng = np.random.default_rng(11)
    for _ in range(60):
        G = int(rng.integers(20, 120))
        pts = np.unique(rng.integers(0, G, size=(int(rng.integers(30, 400)), 2)), axis=0)
        if bo._is_degenerate(pts):
            continue
        ref = oracle_tris(pts)
        for which in ("star_host_triangulate", "star_host_triangulate_local"):
            lib.star_host_use_cache(1)
            assert run(getattr(lib, which), pts, G, G)[0] == ref
    lib.star_host_use_cache(0)


def test_realistic_render_sites(lib):
    hyp = synthetic.make_hypotheses(16, 2, seed=0)
    p0, p1 = synthetic.make_pano(0), synthetic.make_pano(1)
    a = bo.xyzrgb_from_arrays(p0[1], p0[0], bo.floor_ceiling_z_range("floor"))
    a, _ = bo.pose_pair(a, a[:1], hyp.R[0], hyp.t[0])
    res = bo.render_bev_image(a, mode="exact")
    sp, tri = res["site_xy_sorted"], res["tri"]
    ref = sorted(map(tuple, np.sort(sp[tri][:, :, 1] * 4096 + sp[tri][:, :, 0], 1).tolist()))
    lib.star_host_use_table(0)
    assert run(lib.star_host_triangulate_local, sp, 501, 501)[0] == ref
    lib.star_host_use_table(1)
    got, stats = run(lib.star_host_triangulate_local, sp, 501, 501)
    assert got == ref
    assert stats[1] < 0.05 * len(sp)  # only a few per cent of the sites need the general walk
    assert run(lib.star_host_triangulate, sp, 501, 501)[0] == ref


def left_out(lib, pts, H, W):
    xs = np.ascontiguousarray(pts[:, 0], dtype=np.int32)
    ys = np.ascontiguousarray(pts[:, 1], dtype=np.int32)
    stats = np.zeros(4, dtype=np.int64)
    bad = lib.star_host_check_left_out(xs.ctypes.data_as(ctypes.c_void_p), ys.ctypes.data_as(ctypes.c_void_p), len(xs), H, W,
                                       stats.ctypes.data_as(ctypes.c_void_p))
    return bad, int(stats[0]), int(stats[1])


def test_sites_left_out_of_the_list_own_unit_triangles_only(lib):
    """star_local.h sdl_walk_word: a site whose left, right and upper neighbours are sites is not walked by the kernel (until
    round 3: five neighbours).  Every triangle such a site owns must be a unit triangle (nothing to rasterise) -- on random sets
    from half to nearly full occupancy, full lattices, image borders (the bitmap word boundaries at x = 31 / 32 / 63 / 64
    included) and a real render's sites."""
    lib.star_host_use_table(1)
    rng = np.random.default_rng(23)
    seen = 0
    for _ in range(120):
        G = int(rng.integers(8, 100))
        dens = rng.uniform(0.3, 0.98)
        pts = np.argwhere(rng.random((G, G)) < dens)[:, ::-1].copy()
        if len(pts) < 4 or bo._is_degenerate(pts):
            continue
        bad, n_out, owned = left_out(lib, pts, G, G)
        assert bad == 0
        assert owned >= n_out          # at least one owned triangle each: (s, right neighbour, one of the two above)
        seen += n_out
    assert seen > 20000
    yy, xx = np.mgrid[0:40, 0:70]
    full = np.stack([xx.ravel(), yy.ravel()], 1)
    bad, n_out, _ = left_out(lib, full, 40, 70)
    assert bad == 0 and n_out == 39 * 68   # every site with both horizontal neighbours and a row above
    hyp = synthetic.make_hypotheses(16, 2, seed=0)
    p0 = synthetic.make_pano(0)
    a = bo.xyzrgb_from_arrays(p0[1], p0[0], bo.floor_ceiling_z_range("floor"))
    a, _ = bo.pose_pair(a, a[:1], hyp.R[0], hyp.t[0])
    _, img_xy = bo.bev_pixel_indices(a[:, :3])
    sites = np.unique(img_xy, axis=0)
    bad, n_out, _ = left_out(lib, sites, 501, 501)
    print(f"real render: {n_out} of {len(sites)} sites need no walk ({100.0 * n_out / len(sites):.1f} %)")
    assert bad == 0 and n_out > 0.45 * len(sites)


def test_the_walk_counter_tool_builds_and_reproduces_the_triangulation(tmp_path):
    """tools/probe/host/walk_counters.cpp (development: the general walk's counters with the kernel's E2 schedule simulated) compiles against
    the kernel's headers as they are, and its simulated schedule -- eight wavefronts, runs of eight, delayed cache visibility -- emits
    the oracle's triangles for a realistic site set."""
    exe = tmp_path / "walk_counters"
    subprocess.run(["g++", "-O2", "-o", str(exe), str(ROOT / "tools" / "probe" / "host" / "walk_counters.cpp")], check=True)
    hyp = synthetic.make_hypotheses(16, 2, seed=0)
    p0 = synthetic.make_pano(0)
    a = bo.xyzrgb_from_arrays(p0[1], p0[0], bo.floor_ceiling_z_range("floor"))
    a, _ = bo.pose_pair(a, a[:1], hyp.R[3], hyp.t[3])
    res = bo.render_bev_image(a, mode="exact")
    sites = tmp_path / "sites.bin"
    res["site_xy_sorted"].astype(np.int32).tofile(sites)
    out = subprocess.run([str(exe), str(sites)], check=True, capture_output=True, text=True).stdout
    assert f"sites {len(res['site_xy_sorted'])} triangles {len(res['tri'])} " in out, out
