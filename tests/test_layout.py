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
    # integer square root and the distance in 1/256 pixels
    for v in (0, 1, 2, 3, 4, 15, 16, 17, 2 ** 40 + 12345, 2 ** 62 - 1):
        r = lo.isqrt64(v)
        assert r * r <= v < (r + 1) * (r + 1)
    assert lo.segment_distance_256(0, 3, 0, 0, 10, 0) == 768 and lo.segment_distance_256(13, 4, 0, 0, 10, 0) == 1280


def test_thick_line_coverage_and_layout_image():
    img = np.zeros((40, 60, 3), np.uint8)
    lo.thick_line_aa(img, 10, 20, 50, 20, (0, 255, 0), 8)
    col = img[:, 30, 1].astype(int)
    assert (col[17:24] == 255).all() and col[16] == 128 and col[24] == 128 and col[15] == 0 and col[25] == 0   # 8 px wide, soft rim
    assert img[20, 5, 1] > 0 and img[20, 4, 1] == 0 or img[20, 6, 1] > 0                                       # round caps
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
    dev = torch.device("cuda:0")
    got = layout.rasterise_layouts(specs, dev)
    from salve_amd.rasteriser import BevRasteriser

    u8 = BevRasteriser(dev).export_u8(got).cpu().numpy()
    for k, (room, wdos) in enumerate(specs):
        exp = lo.rasterize_single_layout(room, wdos)
        assert np.array_equal(u8[k], exp), f"layout {k}"


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
