"""The rasterised-LAYOUT modality: the oracle's rules on the CPU, and the HIP kernel against the
oracle bit for bit on the GPU.  OpenCV is not installed: the oracle is "parity unpinned" (oracle/layout_oracle.py)."""

from types import SimpleNamespace

import numpy as np
import pytest

from oracle import layout_oracle as lo


def bresenham_reference(x1, y1, x2, y2):
    """OpenCV's LineIterator (8-connected) step by step: err = dx - 2 dy; a minor step whenever err < 0."""
    dx, dy = abs(x2 - x1), abs(y2 - y1)
    sx, sy = (1 if x2 >= x1 else -1), (1 if y2 >= y1 else -1)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err, x, y, out = dx - 2 * dy, x1, y1, []
    for _ in range(dx + 1):
        out.append((x, y))
        minor = err < 0
        err += -2 * dy + (2 * dx if minor else 0)
        if steep:
            y += sy
            x += sx if minor else 0
        else:
            x += sx
            y += sy if minor else 0
    return out


def test_line_membership_closed_form_equals_the_iterator():
    rng = np.random.default_rng(0)
    for _ in range(200):
        x1, y1, x2, y2 = (int(v) for v in rng.integers(-20, 21, size=4))
        pts = set(bresenham_reference(x1, y1, x2, y2))
        got = {(x, y) for x in range(-22, 23) for y in range(-22, 23) if lo.on_line8(x, y, x1, y1, x2, y2)}
        assert got == pts, (x1, y1, x2, y2)


def test_fill_poly_rules():
    img = np.zeros((40, 40, 3), np.uint8)
    lo.fill_poly(img, np.array([[5, 5], [30, 5], [30, 20], [5, 20], [5, 5]]), (255, 255, 255))
    filled = img.any(-1)
    assert filled[5:21, 5:31].all() and filled.sum() == 16 * 26          # axis-aligned rectangle: both borders included
    # a concave polygon: the notch stays empty, the boundary is drawn
    img = np.zeros((60, 60, 3), np.uint8)
    poly = np.array([[5, 5], [50, 5], [50, 50], [28, 20], [5, 50]])
    lo.fill_poly(img, poly, (255, 255, 255))
    assert not img[45, 28].any() and img[10, 28].all() and img[50, 5].all() and img[50, 50].all()


def test_opencv_thick_line_rules():
    """The restated ThickLine (oracle/layout_oracle.py: parity unpinned, OpenCV 4.x drawing.cpp).  Properties any faithful
    restatement has, checked on an axis-aligned 8-pixel line: the quadrilateral p +- (0, 4 px) fills rows 16 .. 24 completely
    between the end points (FillConvexPoly fills ceil(left) .. floor(right) of the scanlines ymin .. ymax inclusive), its
    anti-aliased edges leave a soft pixel row above and below, the end caps are 12-gons of radius 4 -- rounded, symmetric --
    and the geometry is exactly ThickLine's: dp = cvRound(perpendicular * 4 px) in 16.16 fixed point."""
    quad, th, p0, p1 = lo.thick_line_geometry(10, 20, 50, 20, 8)
    one = 1 << 16
    assert th == 4 * one and p0 == (10 * one, 20 * one) and p1 == (50 * one, 20 * one)
    assert quad == [(10 * one, 16 * one), (10 * one, 24 * one), (50 * one, 24 * one), (50 * one, 16 * one)]   # dp = (0, -4 px): dx = p0.x - p1.x < 0
    cap = lo.cv_ellipse_poly(100 * one, 100 * one, 4 * one)
    assert len(cap) == 13 and cap[0] == cap[12] == (104 * one, 100 * one) and cap[3] == (100 * one, 104 * one)
    assert cap[1] == (100 * one + 227023, 102 * one) and cap[2] == (102 * one, 100 * one + 227023)   # 4 * 0.8660254 = 3.4641016 px
    img = np.zeros((40, 60, 3), np.uint8)
    lo.cv_thick_line_aa(img, 10, 20, 50, 20, (0, 255, 0), 8)
    col = img[:, 30, 1].astype(int)
    assert (col[16:25] == 255).all() and col[14] == 0 and col[26] == 0 and 0 < col[15] < 128 and 0 < col[25] < 128   # (the filter's two tails differ: 48 and 53 at dist 16)
    row = img[20, :, 1].astype(int)
    assert (row[6:55] == 255).all() and row[4] == 0 and row[56] == 0                 # caps reach 4 px beyond the end points
    assert 380 <= int((img[..., 1] == 255).sum()) <= 470                             # a 40 x 9 bar + two discs of radius 4
    assert not img[..., 0].any() and not img[..., 2].any()
    # a zero-length segment draws its two (identical) caps only; thickness 1 is a single LineAA
    dot = np.zeros((30, 30, 3), np.uint8)
    lo.cv_thick_line_aa(dot, 15, 15, 15, 15, (255, 255, 255), 8)
    assert dot[15, 15].all() and dot[15, 11].all() and dot[15, 19].all() and not dot[15, 9].any() and not dot[9, 15].any()
    thin = np.zeros((30, 30, 3), np.uint8)
    lo.cv_thick_line_aa(thin, 5, 5, 25, 12, (255, 255, 255), 1)
    assert 20 < (thin[..., 0] > 0).sum() < 80 and thin.max() <= 255


def test_clip_line_fixed_point():
    W = H = 100 << 16
    assert lo.clip_line_fixed(W, H, 5 << 16, 5 << 16, 50 << 16, 60 << 16) == (5 << 16, 5 << 16, 50 << 16, 60 << 16)
    assert lo.clip_line_fixed(W, H, -(10 << 16), -(10 << 16), -(5 << 16), 50 << 16) is None
    x1, y1, x2, y2 = lo.clip_line_fixed(W, H, -(10 << 16), 50 << 16, 200 << 16, 50 << 16)
    assert (x1, y1, x2, y2) == (0, 50 << 16, W - 1, 50 << 16)


def test_layout_image():
    room = np.array([[-1.5, -1.0], [2.0, -1.2], [2.2, 0.5], [0.8, 0.6], [0.7, 1.9], [-1.4, 1.8]])
    wdos = [("doors", np.array([[2.0, -1.2], [2.1, -0.4]])), ("windows", np.array([[-1.5, -0.5], [-1.45, 0.6]]))]
    out = lo.rasterize_single_layout(room, wdos)
    assert out.shape == (501, 501, 3) and out.dtype == np.uint8
    px = lo.to_pixels(np.array([[0.0, 0.0]]) * 1.5)[0]
    assert (out[500 - px[1], px[0]] == 255).all()                          # the room's inside is white, image flipped
    assert ((out[..., 1] == 255) & (out[..., 0] == 0)).sum() > 200         # a pure green door


gpu = pytest.mark.gpu


@gpu
def test_layout_kernel_matches_oracle_bit_for_bit():
    torch = pytest.importorskip("torch")
    from salve_amd import layout
    from salve_amd.common.sim2 import Sim2

    rng = np.random.default_rng(3)
    specs = []
    for k in range(6):
        n = int(rng.integers(4, 12))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(0.8, 3.2, n)
        room = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1) + rng.uniform(-1, 1, 2)
        wdos = []
        for j in range(int(rng.integers(0, 6))):
            a = int(rng.integers(0, n))
            p, q = room[a], room[(a + 1) % n]
            t0, t1 = np.sort(rng.uniform(0, 1, 2))
            wdos.append((("doors", "windows", "openings")[j % 3], np.stack([p + t0 * (q - p), p + t1 * (q - p)])))
        specs.append((np.vstack([room, room[:1]]), wdos))
    specs.append((np.array([[-9.0, -9.0], [9.0, -9.0], [9.0, 9.0], [-9.0, 9.0]]), [("doors", np.array([[-20.0, 0.0], [20.0, 0.3]]))]))  # beyond the image
    # seventeen segments in one image (the kernel sets its primitives up in chunks of seven), crossing one another -- the blends
    # are order dependent --, one of zero length (end caps only), one leaving the image through a corner
    star = [(("doors", "windows", "openings")[j % 3], np.array([[0.3 * np.cos(j), 0.3 * np.sin(j)], [2.5 * np.cos(j * 0.7), 2.5 * np.sin(j * 0.7)]])) for j in range(15)]
    star += [("doors", np.array([[1.0, 1.0], [1.0, 1.0]])), ("windows", np.array([[2.9, 2.9], [4.5, 4.4]]))]
    specs.append((np.array([[-3.0, -3.0], [3.0, -3.0], [3.0, 3.0], [-3.0, 3.0], [-3.0, -3.0]]), star))
    dev = torch.device("cuda:0")
    from salve_amd.rasteriser import BevRasteriser

    for render_mask in (True, False):   # False: the room as a 2-pixel anti-aliased contour instead of the filled mask (:128-136)
        got = layout.rasterise_layouts(specs, dev, render_mask=render_mask)
        u8 = BevRasteriser(dev).export_u8(got).cpu().numpy()
        for k, (room, wdos) in enumerate(specs):
            exp = lo.rasterize_single_layout(room, wdos, render_mask=render_mask)
            assert np.array_equal(u8[k], exp), f"layout {k}, render_mask {render_mask}: {int((u8[k] != exp).any(-1).sum())} pixels differ"


@gpu
def test_a_segment_too_thick_for_the_kernel_is_left_out_and_reported():
    """Through the C ABI directly (what INTEGRATION.md advertises): a segment of 19 pixels or more -- OpenCV would give it end caps at
    18- / 5-degree steps, which the kernel does not draw -- is NOT drawn and SALVE_STATUS_LAYOUT_THICKNESS is raised; the 8-pixel
    segment next to it is drawn as always.  (The Python wrapper refuses such thicknesses before they reach the library.)"""
    import ctypes

    torch = pytest.importorskip("torch")
    from salve_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    H = W = 64
    rec = np.zeros(2, dtype=_lib.LAYOUT_DTYPE)
    rec[0]["n_seg"], rec[0]["seg_off"] = 1, 0     # image 0: one 8-pixel segment
    rec[1]["n_seg"], rec[1]["seg_off"] = 2, 0     # image 1: the same segment + a 19-pixel one across it
    segs = np.array([[10, 20, 50, 24, 0x00FF00, 8, 0, 0], [12, 50, 52, 40, 0x0000FF, 19, 0, 0]], dtype=np.int32)
    d_rec = torch.from_numpy(rec.view(np.uint8)).to(dev)
    d_seg = torch.from_numpy(segs).to(dev)
    d_poly = torch.zeros((1, 2), dtype=torch.int32, device=dev)
    out = torch.empty((2, H, W), dtype=torch.int32, device=dev)
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    st = lib.salve_layout_rasterise(ctypes.c_void_p(d_rec.data_ptr()), 2, ctypes.c_void_p(d_poly.data_ptr()), ctypes.c_void_p(d_seg.data_ptr()), H, W,
                                    ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(word.data_ptr()), None)
    torch.cuda.synchronize()
    assert st == 0, lib.salve_last_error()
    assert int(word.item()) == _lib.STATUS_LAYOUT_THICKNESS
    assert out[0].any() and torch.equal(out[0], out[1])     # the thick segment left no pixel; the thin one is there
    word.zero_()
    st = lib.salve_layout_rasterise(ctypes.c_void_p(d_rec.data_ptr()), 1, ctypes.c_void_p(d_poly.data_ptr()), ctypes.c_void_p(d_seg.data_ptr()), H, W,
                                    ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(word.data_ptr()), None)
    torch.cuda.synchronize()
    assert st == 0 and int(word.item()) == 0


@gpu
def test_layout_facade_pair_and_files(tmp_path):
    torch = pytest.importorskip("torch")
    from salve_amd.common.sim2 import Sim2
    from salve_amd.utils import bev_rendering_utils as bru
    from salve_amd.utils import image_io

    wdo = lambda t, a, b: SimpleNamespace(type=t, vertices_local_2d=np.array([a, b], dtype=np.float64))
    node = lambda verts, doors, windows, openings: SimpleNamespace(room_vertices_local_2d=np.array(verts, dtype=np.float64), doors=doors, windows=windows, openings=openings)
    graph = SimpleNamespace(nodes={
        4: node([[-1, -1], [1.5, -1], [1.5, 1.2], [-1, 1.2]], [wdo("doors", [1.5, -0.5], [1.5, 0.3])], [wdo("windows", [-1, 0], [-1, 0.8])], []),
        7: node([[-2, -1.5], [1, -1.5], [1, 1], [-2, 1]], [], [], [wdo("openings", [1, -1], [1, 0.2])]),
    })
    th = np.deg2rad(30.0)
    S = Sim2(np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]), np.array([0.4, -0.3]), 1.0)
    img1, img2 = bru.rasterize_room_layout_pair(S, graph, "0001", "floor_01", 4, 7)
    posed_room = S.transform_from(np.vstack([graph.nodes[4].room_vertices_local_2d, graph.nodes[4].room_vertices_local_2d[:1]]))
    exp1 = lo.rasterize_single_layout(posed_room, [(w.type, S.transform_from(w.vertices_local_2d)) for w in graph.nodes[4].doors + graph.nodes[4].windows])
    exp2 = lo.rasterize_single_layout(np.vstack([graph.nodes[7].room_vertices_local_2d, graph.nodes[7].room_vertices_local_2d[:1]]),
                                      [(w.type, w.vertices_local_2d) for w in graph.nodes[7].openings])
    assert np.array_equal(img1, exp1) and np.array_equal(img2, exp2)
    # generate_texture_maps_for_pair with the layout modality writes the two tiles under layout_save_root, floor names only
    pair = tmp_path / "hyp" / "4_7__door_0_0_identity.json"
    pair.parent.mkdir(parents=True)
    S.save_as_json(str(pair))
    fpaths = {4: "/z/0001/panos/floor_01_partial_room_01_pano_4.jpg", 7: "/z/0001/panos/floor_01_partial_room_02_pano_7.jpg"}
    for surface in ("floor", "ceiling"):
        bru.generate_texture_maps_for_pair(fpaths, surface, str(pair), 3, "gt_alignment_approx", str(tmp_path / "bev"), "0001", "floor_01",
                                           str(tmp_path / "depth"), ["layout"], str(tmp_path / "layout"), graph)
    files = sorted(p.name for p in (tmp_path / "layout" / "gt_alignment_approx" / "0001").glob("*.jpg"))
    assert files == ["pair_3___door_0_0_identity_floor_rgb_floor_01_partial_room_01_pano_4.jpg",
                     "pair_3___door_0_0_identity_floor_rgb_floor_01_partial_room_02_pano_7.jpg"]
    back = image_io.read_rgb(str(tmp_path / "layout" / "gt_alignment_approx" / "0001" / files[0]))
    assert back.shape == (501, 501, 3) and np.abs(back.astype(int) - img1.astype(int)).mean() < 4   # JPEG is lossy
