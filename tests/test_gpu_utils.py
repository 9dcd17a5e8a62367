"""GPU tests of the stand-alone utilities (z-order, interpolation, mask) and of the 2048x1024 / ResNet-152 4-tuple config."""

from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import bev_oracle as bo  # noqa: E402
from oracle import resnet_oracle as ro  # noqa: E402
from salve_amd import synthetic  # noqa: E402
from salve_amd.utils import interpolation_utils as iu  # noqa: E402
from salve_amd.utils import zorder_utils as zu  # noqa: E402


def test_zorder_reference_kats_and_golden(golden_dir):
    # reference tests/utils/test_zorder_utils.py:8-66
    cases = [
        ([[0, 1, 0], [1, 2, 4], [0, 1, 5], [5, 6, 1]], dict(zmin=0, zmax=10, num_slices=5), [False, True, True, True]),
        ([[0, 1, 0], [1, 2, 4], [2, 3, 5], [3, 4, 1]], dict(zmin=0, zmax=10, num_slices=5), [True, True, True, True]),
        ([[0, 1, 0], [0, 1, 1], [0, 1, 2], [0, 1, 3]], dict(zmin=0, zmax=10, num_slices=5), [False, False, False, True]),
        ([[0, 1, 0], [0, 1, 1], [0, 1, 10], [0, 1, 11]], dict(zmin=0, zmax=10, num_slices=5), [False, True, False, False]),
        ([[0, 1, 0], [0, 1, 1], [0, 1, 2], [0, 1, 3]], dict(zmin=0, zmax=4, num_slices=2), [False, False, False, True]),
    ]
    for xyz, kw, exp in cases:
        a = np.array(xyz)
        assert zu.choose_elevated_repeated_vals(a[:, 0], a[:, 1], a[:, 2], **kw).tolist() == exp
    g = np.load(golden_dir / "g2_zorder.npz")
    for c in range(4):
        assert np.array_equal(zu.choose_elevated_repeated_vals(g[f"x{c}"], g[f"y{c}"], g[f"z{c}"]), g[f"valid{c}"])


def test_remove_hallucinated_content_kat_and_golden(golden_dir):
    s = np.zeros((6, 6), dtype=np.int64)
    s[0, 1], s[0, 3], s[2, 1], s[4, 1] = 2, 4, 2, 2
    sparse = np.stack([s, s, s], -1)
    interp = np.stack([np.tile(np.arange(1, 7), (6, 1))] * 3, -1)
    out = iu.remove_hallucinated_content(sparse, interp, K=3)
    exp = np.array([[1, 2, 3, 4, 5, 0], [1, 2, 3, 4, 5, 0], [1, 2, 3, 0, 0, 0], [1, 2, 3, 0, 0, 0], [1, 2, 3, 0, 0, 0],
                    [1, 2, 3, 0, 0, 0]], dtype=np.uint8)
    for c in range(3):
        assert np.array_equal(out[:, :, c], exp)
    g = np.load(golden_dir / "g3_mask.npz")
    assert np.array_equal(iu.remove_hallucinated_content(g["sparse"], g["interp"]), g["out"])


def test_interp_dense_grid_from_sparse():
    # reference tests/utils/test_interpolation_utils.py:8-82
    col = np.array([[255, 0, 0], [0, 255, 0], [255, 0, 0], [0, 255, 0]])
    for pts in ([[0, 0], [0, 3], [0, 2], [0, 4]], [[0, 0], [3, 0], [2, 0], [4, 0]], [[1, 1], [5, 5]]):
        img = np.zeros((10, 10, 3))
        out = iu.interp_dense_grid_from_sparse(img, np.array(pts), col[: len(pts)], 10, 10, False)
        assert np.allclose(out, np.zeros((10, 10, 3)))
    img = np.zeros((4, 4, 3))
    out = iu.interp_dense_grid_from_sparse(img, np.array([[0, 0], [0, 3], [3, 3], [3, 0]]),
                                           np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 0, 0]]), 4, 4, False)
    assert isinstance(out, np.ndarray) and out.shape == (4, 4, 3)
    rng = np.random.default_rng(0)
    for G in (40, 131):
        pts = np.unique(rng.integers(0, G, size=(G * 6, 2)), axis=0)
        colr = rng.integers(0, 256, size=(pts.shape[0], 3)).astype(np.float64)
        got = iu.interp_dense_grid_from_sparse(np.zeros((G, G, 3), dtype=np.uint8), pts, colr, G, G, False)
        assert np.array_equal(got, bo.interp_exact(pts, colr.astype(np.uint8), G, G)[0])


def test_interp_of_a_half_occupied_image():
    """A 501 x 501 image with every second pixel a site (white noise): 97 k triangles of twice-the-area 2 against a midpoint
    queue of 83 k entries (bev_render.hip: QueueEmit) -- the queue-full path, which no render of a texture map reaches.  Also
    35 % occupancy: the general-triangle queue at 73 % of its capacity."""
    G = 501
    rng = np.random.default_rng(3)
    for occ in (0.5, 0.35):
        ys, xs = np.nonzero(rng.random((G, G)) < occ)
        pts = np.stack([xs, ys], 1)
        colr = rng.integers(0, 256, size=(pts.shape[0], 3)).astype(np.float64)
        got = iu.interp_dense_grid_from_sparse(np.zeros((G, G, 3), dtype=np.uint8), pts, colr, G, G, False)
        assert np.array_equal(got, bo.interp_exact(pts, colr.astype(np.uint8), G, G)[0]), occ


def test_triangle_queue_at_its_bound_on_regular_half_occupied_lattices():
    """The ONE triangle queue of the densify kernel holds H W entries; its bound (every queued triangle holds a lattice point that
    is not a site, so there are at most 2 (H W - n) of them, and 2 n triangles in all) is reached by REGULAR half-occupied
    lattices: a checkerboard, every other column, every other row -- 125 k sites, (H - 1)(W - 1) = 250 000 triangles of
    twice-the-area 2, every one queued, maximal co-circularity (each star resolves its cocircular quads by the perturbation on
    its own).  Through the C ABI with the work counters: queued triangles <= H W, no status bit, and the interpolant is the
    oracle's, bit for bit."""
    import ctypes

    from salve_amd import _lib, status
    from salve_amd.common.bevparams import BEVParams
    from salve_amd.rasteriser import BevRasteriser

    G = 501
    dev = torch.device("cuda:0")
    ras = BevRasteriser(dev, bev_params=BEVParams(img_h=G - 1, img_w=G - 1, meters_per_px=1.0))
    ras.cfg.out_flags = 3   # no flip, no mask: the plain interpolant
    yy, xx = np.mgrid[0:G, 0:G]
    status.check(dev, "before")
    for name, m in (("checkerboard", (xx + yy) % 2 == 0), ("every other column", xx % 2 == 0), ("every other row", yy % 2 == 0)):
        pts = np.stack([xx[m], yy[m]], 1)
        col = np.stack([(xx[m] * 7 + yy[m] * 3) % 256, (xx[m] * 5 + 11) % 256, (yy[m] * 13) % 256], 1).astype(np.uint8)
        xy = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.int32)).to(dev)
        rgb = torch.from_numpy(col).to(dev)
        bev = torch.empty((1, G, G), dtype=torch.int32, device=dev)
        ras.keys_from_pixels(xy, rgb, bev)
        stats = torch.zeros((1, 8), dtype=torch.int32, device=dev)
        ws = ras._workspace(1)
        st = ras.lib.salve_bev_densify(ctypes.byref(ras.cfg), 1, ctypes.c_void_p(bev.data_ptr()), None, ctypes.c_void_p(stats.data_ptr()),
                                       status.ptr(dev), ctypes.c_void_p(ws.data_ptr()), ws.numel(), None)
        assert st == 0, ras.lib.salve_last_error()
        torch.cuda.synchronize()
        status.check(dev, name)      # raises on SALVE_STATUS_WALK_FAILED (a full queue sets it)
        sv = stats.cpu().numpy()[0]
        assert sv[0] == len(pts) and sv[5] == 0
        assert 0 < sv[7] <= G * G, (name, sv)
        print(f"{name}: {sv[0]} sites, {sv[7]} queued triangles of {G * G} queue entries, {sv[6]} hard sites")
        got = ras.export_u8(bev)[0].cpu().numpy()
        assert np.array_equal(got, bo.interp_exact(pts, col, G, G)[0]), name


def test_config5_large_panos_two_surfaces_resnet152():
    """2048x1024 panoramas, floor + ceiling, ResNet-152 12-channel early fusion (BASELINE config 5, fp16)."""
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline
    from _helpers import randomise_bn

    dev = torch.device("cuda:0")
    H, W = 1024, 2048
    panos = [synthetic.make_pano(i, H, W) for i in range(2)]
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(152, False, 2, SimpleNamespace(modalities=["ceiling_rgb_texture", "floor_rgb_texture"]))
    randomise_bn(model)
    model.eval()
    pipe = RenderVerifyPipeline(model, dev, pano_hw=(H, W), chunk=4)
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    hyp = synthetic.make_hypotheses(3, 2, seed=2)
    prepared = pipe.prepare(hyp)
    logits = pipe.score(prepared).cpu()
    torch.cuda.synchronize()
    ck, k0 = pipe.bev_index(prepared, 0)
    bev = pipe.ras.export_u8(pipe.bevs[pipe.last_chunk_buffer[ck]][k0:k0 + 2]).cpu().numpy()  # hypothesis 0: ceiling, floor of pano i1
    tiles = []
    for si, surface in enumerate(("ceiling", "floor")):
        i1, i2 = int(hyp.i1[0]), int(hyp.i2[0])
        r1, r2 = bo.render_bev_pair(panos[i1][0], panos[i1][1], panos[i2][0], panos[i2][1], hyp.R[0], hyp.t[0], surface, mode="exact")
        assert np.array_equal(bev[si], r1["bev"]), surface
        tiles += [bo.tile_from_bev(r1["bev"]), bo.tile_from_bev(r2["bev"])]
    got_tiles = pipe.tiles[0].float().cpu().permute(2, 0, 1)
    exp_tiles = torch.from_numpy(np.concatenate(tiles, 0))
    assert torch.equal(got_tiles[:12], exp_tiles.half().float()) and not got_tiles[12:].any()
    with torch.no_grad():
        ref = ro.forward(model.state_dict(), 152, [t[None] for t in exp_tiles.split(3)])  # fp32 tiles, unquantised
    err = float((logits[:1] - ref).abs().max())
    print(f"config 5: logits {logits[0].tolist()} oracle {ref[0].tolist()} err {err:.2e}")
    assert err <= 1e-3  # north_star: logits within 1e-3, absolute
    pipe.check("config 5")
