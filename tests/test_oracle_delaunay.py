"""Properties of the canonical exact Delaunay restatement (oracle/csrc/delaunay_exact.c)."""

import numpy as np

from oracle import bev_oracle as bo


def _random_sites(rng, n, g):
    return np.unique(rng.integers(0, g, size=(n, 2)), axis=0)


def test_triangulations_are_delaunay_and_cover_the_hull():
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(1)
    done = 0
    for _ in range(300):
        pts = _random_sites(rng, int(rng.integers(3, 80)), int(rng.integers(3, 14)))
        if bo._is_degenerate(pts):
            continue
        order, tri = bo.delaunay_exact(pts[:, 0], pts[:, 1])
        sp = pts[order]
        if tri.shape[0] == 0:  # all collinear on a diagonal
            continue
        assert bo.check_delaunay(sp, tri)
        a = sp[tri[:, 0]].astype(np.int64)
        b = sp[tri[:, 1]].astype(np.int64)
        c = sp[tri[:, 2]].astype(np.int64)
        area2 = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]))
        assert (area2 > 0).all()
        assert abs(area2.sum() / 2 - ConvexHull(sp).volume) < 1e-9
        done += 1
    assert done > 100


def test_canonical_triangulation_ignores_input_order():
    """The perturbation rank is the raster order of the site, not the position in the input."""
    rng = np.random.default_rng(2)
    pts = _random_sites(rng, 400, 24)  # dense lattice: co-circular quadruples everywhere
    col = rng.integers(0, 256, size=(pts.shape[0], 3), dtype=np.uint8)
    base = bo.interp_exact(pts, col, 24, 24)
    for _ in range(5):
        perm = rng.permutation(pts.shape[0])
        got = bo.interp_exact(pts[perm], col[perm], 24, 24)
        assert np.array_equal(got[0], base[0])
        assert np.array_equal(got[3], base[3])


def test_exact_equals_scipy_where_the_triangulation_is_unique():
    rng = np.random.default_rng(3)
    for _ in range(20):
        pts = _random_sites(rng, 60, 40)
        col = rng.integers(0, 256, size=(pts.shape[0], 3), dtype=np.uint8)
        u8, f, cover, tri, order = bo.interp_exact(pts, col, 40, 40)
        _, fs = bo.interp_scipy(pts, col.astype(float), 40, 40)
        assert np.array_equal(np.isfinite(fs).all(-1), cover)
        strong = bo.strongly_delaunay_pixels(pts[order], tri, 40, 40)
        if strong.any():
            assert np.abs(f[strong] - fs[strong]).max() < 1e-9


def test_interpolation_is_exact_on_affine_colour_fields():
    """Linear interpolation reproduces an affine field exactly; floor of the exact rational is the field."""
    rng = np.random.default_rng(4)
    pts = _random_sites(rng, 300, 50)
    val = (2 * pts[:, 0] + 3 * pts[:, 1]).astype(np.uint8)  # <= 250
    col = np.stack([val, val, val], -1)
    u8, f, cover, _, _ = bo.interp_exact(pts, col, 50, 50)
    yy, xx = np.nonzero(cover)
    assert np.array_equal(u8[yy, xx, 0], (2 * xx + 3 * yy).astype(np.uint8))
