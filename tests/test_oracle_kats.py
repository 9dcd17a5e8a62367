"""The reference's own known-answer tests for the hot path, re-expressed on the oracle.

Each test cites the reference test it restates (paths relative to /root/reference); the
expected values are the reference's.
"""

import numpy as np

from oracle import bev_oracle as bo


# tests/utils/test_zorder_utils.py:8-66
def _zorder(xyz, **kw):
    xyz = np.array(xyz)
    return bo.choose_elevated(xyz[:, 0], xyz[:, 1], xyz[:, 2], **kw)


def test_zorder_single_repeat():
    v = _zorder([[0, 1, 0], [1, 2, 4], [0, 1, 5], [5, 6, 1]], zmin=0, zmax=10, num_slices=5)
    assert v.tolist() == [False, True, True, True]


def test_zorder_no_repeats():
    v = _zorder([[0, 1, 0], [1, 2, 4], [2, 3, 5], [3, 4, 1]], zmin=0, zmax=10, num_slices=5)
    assert v.tolist() == [True, True, True, True]


def test_zorder_all_repeated():
    v = _zorder([[0, 1, 0], [0, 1, 1], [0, 1, 2], [0, 1, 3]], zmin=0, zmax=10, num_slices=5)
    assert v.tolist() == [False, False, False, True]


def test_zorder_out_of_range_z():
    # z = 10 is excluded: the upper boundary is exclusive
    v = _zorder([[0, 1, 0], [0, 1, 1], [0, 1, 10], [0, 1, 11]], zmin=0, zmax=10, num_slices=5)
    assert v.tolist() == [False, True, False, False]


def test_zorder_two_slices():
    v = _zorder([[0, 1, 0], [0, 1, 1], [0, 1, 2], [0, 1, 3]], zmin=0, zmax=4, num_slices=2)
    assert v.tolist() == [False, False, False, True]


# tests/utils/test_interpolation_utils.py:8-59
def test_interp_collinear_and_too_few_points_give_zeros():
    col = np.array([[255, 0, 0], [0, 255, 0], [255, 0, 0], [0, 255, 0]], dtype=np.uint8)
    for pts in ([[0, 0], [0, 3], [0, 2], [0, 4]], [[0, 0], [3, 0], [2, 0], [4, 0]]):
        for fn in (lambda p: bo.interp_exact(p, col, 10, 10)[0], lambda p: bo.interp_scipy(p, col.astype(float), 10, 10)[0]):
            assert not fn(np.array(pts)).any()
    pts = np.array([[1, 1], [5, 5]])
    assert not bo.interp_exact(pts, col[:2], 10, 10)[0].any()
    assert not bo.interp_scipy(pts, col[:2].astype(float), 10, 10)[0].any()


def test_interp_4x4_smoke():
    pts = np.array([[0, 0], [0, 3], [3, 3], [3, 0]])
    col = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 0, 0]], dtype=np.uint8)
    out = bo.interp_exact(pts, col, 4, 4)[0]
    assert out.shape == (4, 4, 3) and out.dtype == np.uint8
    assert np.array_equal(out[0, 0], [255, 0, 0]) and np.array_equal(out[3, 3], [0, 0, 255])


# tests/utils/test_interpolation_utils.py:85-127 (exact 6x6 result for K = 3)
def test_remove_hallucinated_content_kat():
    s = np.zeros((6, 6), dtype=np.int64)
    s[0, 1], s[0, 3], s[2, 1], s[4, 1] = 2, 4, 2, 2
    sparse = np.stack([s, s, s], -1)
    interp = np.stack([np.tile(np.arange(1, 7), (6, 1))] * 3, -1)
    out = bo.remove_hallucinated(sparse, interp, K=3)
    exp = np.array([[1, 2, 3, 4, 5, 0], [1, 2, 3, 4, 5, 0], [1, 2, 3, 0, 0, 0], [1, 2, 3, 0, 0, 0],
                    [1, 2, 3, 0, 0, 0], [1, 2, 3, 0, 0, 0]], dtype=np.uint8)
    for c in range(3):
        assert np.array_equal(out[:, :, c], exp)


# tests/test_hohonet_pano_utils.py:8-24
def test_sphere_table_directions():
    t = bo.sphere_table(512, 1024)
    assert np.allclose(t[0, 0], [0, 0, 1], atol=4e-3)
    assert np.allclose(t[0, 1023], [0, 0, 1], atol=4e-3)
    assert np.allclose(t[511, 0], [0, 0, -1], atol=4e-3)
    assert np.allclose(t[256, 512], [-1, 0, 0], atol=4e-3)


# tests/common/test_bevparams.py:9-35
def test_bev_grid_transform():
    g = bo.BevGrid(img_h=20, img_w=20, meters_per_px=0.5)
    kept, img = bo.bev_pixel_indices(np.array([[2.0, 2, 0], [-5, -5, 0], [5, 5, 0]]), g)
    assert kept.all()
    assert img.tolist() == [[14, 14], [0, 0], [20, 20]]


# tests/utils/test_bev_rendering_utils.py:8-40 (inclusive bounds of prune_to_2d_bbox)
def test_prune_inclusive_bounds():
    g = bo.BevGrid(img_h=2, img_w=2, meters_per_px=1.0)  # window [-1, 1]^2
    pts = np.array([[-2.0, 1.0, 0], [1.0, 0.0, 0], [1.0, 1.0, 0], [0.0, -1.0, 0], [1.0000001, 0, 0]])
    kept, _ = bo.bev_pixel_indices(pts, g)
    assert kept.tolist() == [False, True, True, True, False]
