"""GPU tests of the reference-shaped API (salve_amd.utils.bev_rendering_utils, pipeline) and full-size properties."""

from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import bev_oracle as bo  # noqa: E402
from salve_amd import synthetic  # noqa: E402
from salve_amd.common.bevparams import BEVParams  # noqa: E402
from salve_amd.common.sim2 import Sim2  # noqa: E402
from salve_amd.utils import bev_rendering_utils as bru  # noqa: E402
from salve_amd.utils import image_io  # noqa: E402


@pytest.fixture(scope="module")
def panos():
    return [synthetic.make_pano(i) for i in range(2)]


def posed_cloud(pano, surface, R, t):
    a = bo.xyzrgb_from_arrays(pano[1], pano[0], bo.floor_ceiling_z_range(surface))
    a, _ = bo.pose_pair(a, a[:1], R, t)
    return a


def test_render_bev_image_from_point_cloud(panos):
    hyp = synthetic.make_hypotheses(4, 2, seed=0)
    for hi, surface in ((0, "floor"), (1, "ceiling")):
        cloud = posed_cloud(panos[0], surface, hyp.R[hi], hyp.t[hi])
        got = bru.render_bev_image(BEVParams(), cloud, is_semantics=False)
        assert got.dtype == np.uint8 and got.shape == (501, 501, 3)
        assert np.array_equal(got, bo.render_bev_image(cloud, mode="exact")["bev"])
    far = cloud.copy()
    far[:, :2] += 100.0
    assert bru.render_bev_image(BEVParams(), far, False) is None  # no point in the window -> None (:279)
    with pytest.raises(NotImplementedError):
        bru.render_bev_image(BEVParams(), cloud, True)


def test_render_bev_image_small_geometry():
    rgb, depth = synthetic.make_pano(3, 64, 128)
    hyp = synthetic.make_hypotheses(16, 1, seed=0)
    grid = bo.BevGrid(100, 100, 0.1)
    for hi in (0, 5, 9):
        a = bo.xyzrgb_from_arrays(depth, rgb, bo.floor_ceiling_z_range("floor"))
        a, _ = bo.pose_pair(a, a[:1], hyp.R[hi], hyp.t[hi])
        exp = bo.render_bev_image(a, grid, mode="exact")
        got = bru.render_bev_image(BEVParams(100, 100, 0.1), a, False)
        if exp is None:
            assert got is None
        else:
            assert np.array_equal(got, exp["bev"])


def test_render_bev_pair_and_texture_map_files(panos, tmp_path):
    from PIL import Image

    raw = tmp_path / "raw" / "0001" / "panos"
    raw.mkdir(parents=True)
    depth_root = tmp_path / "depth"
    paths = {}
    for i, (rgb, d) in enumerate(panos):
        p = raw / f"floor_01_partial_room_0{i}_pano_{i}.png"  # lossless stand-in for the panorama JPEG
        Image.fromarray(rgb).save(p)
        image_io.write_depth_png(str(depth_root / "0001" / f"{p.stem}.depth.png"), d)
        paths[i] = str(p)
    hyp = synthetic.make_hypotheses(2, 2, seed=0)
    S = Sim2(hyp.R[0].astype(np.float64), hyp.t[0].astype(np.float64), 1.0)
    args = SimpleNamespace(img_i1=paths[0], img_i2=paths[1], depth_i1=str(depth_root / "0001" / f"{raw.joinpath(paths[0]).stem}.depth.png"),
                           depth_i2=str(depth_root / "0001" / f"{raw.joinpath(paths[1]).stem}.depth.png"), scale=0.001,
                           crop_ratio=80 / 512, crop_z_range=[-float("inf"), -1.0])
    img1, img2 = bru.render_bev_pair(args, "0001", "floor_01", 0, 1, S, False)
    r1, r2 = bo.render_bev_pair(panos[0][0], panos[0][1], panos[1][0], panos[1][1], hyp.R[0], hyp.t[0], "floor", mode="exact")
    assert np.array_equal(img1, r1["bev"]) and np.array_equal(img2, r2["bev"])
    with pytest.raises(ValueError):
        bru.render_bev_pair(SimpleNamespace(scale=0.001), "0001", "floor_01", 0, 1, S, False)
    # the per-pair driver: names, files, idempotent restart
    pair = tmp_path / "hyp" / "0_1__door_0_0_identity.json"
    S.save_as_json(pair)
    bev_root = tmp_path / "bev"
    call = (paths, "floor", str(pair), 7, "gt_alignment_approx", str(bev_root), "0001", "floor_01", str(depth_root), ["rgb_texture"], None, None)
    bru.generate_texture_maps_for_pair(*call)
    out = sorted(p.name for p in (bev_root / "gt_alignment_approx" / "0001").iterdir())
    assert out == ["pair_7___door_0_0_identity_floor_rgb_floor_01_partial_room_00_pano_0.jpg",
                   "pair_7___door_0_0_identity_floor_rgb_floor_01_partial_room_01_pano_1.jpg"]
    first = (bev_root / "gt_alignment_approx" / "0001" / out[0])
    mtime = first.stat().st_mtime_ns
    bru.generate_texture_maps_for_pair(*call)
    assert first.stat().st_mtime_ns == mtime
    back = np.asarray(Image.open(first))
    assert back.shape == (501, 501, 3) and np.abs(back.astype(int) - img1.astype(int)).mean() < 6  # JPEG is lossy


def test_full_size_properties():
    """Size-independent properties on the benchmark-size workload (no oracle needed)."""
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    dev = torch.device("cuda:0")
    P, N = 8, 96
    panos = [synthetic.make_pano(i) for i in range(P)]
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    pipe = RenderVerifyPipeline(model, dev, chunk=16)  # three HIP streams (default): scatter | densify | verify, 6 chunks
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    table = synthetic.make_hypotheses(N, P, seed=1)
    # identity pose through the posed branch == the cached identity render
    table.R[0] = np.eye(2, dtype=np.float32)
    table.t[0] = 0
    prep = pipe.prepare(table)
    a = pipe.score(prep).clone()
    b = pipe.score(prep)
    torch.cuda.synchronize()
    assert torch.equal(a, b)  # deterministic
    assert torch.isfinite(a).all()
    pipe2 = RenderVerifyPipeline(model, dev, chunk=32, overlap=False)  # one stream, other batching
    pipe2.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    c = pipe2.score(pipe2.prepare(table))
    torch.cuda.synchronize()
    assert torch.equal(a, c)  # independent of the batching
    pipe3 = RenderVerifyPipeline(model, dev, chunk=40, streams=2)  # two streams: rasteriser | verifier
    pipe3.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    assert torch.equal(a, pipe3.score(pipe3.prepare(table)))
    prep0 = pipe.prepare(table.shard(0, 6))      # the first 16 hypotheses: one chunk
    pipe.score(prep0)
    torch.cuda.synchronize()
    ck, k0 = pipe.bev_index(prep0, 0)            # where hypothesis 0's render landed (renders run in panorama order)
    assert torch.equal(pipe.bevs[pipe.last_chunk_buffer[ck]][k0], pipe.ref_bev[int(table.i1[0])])
    # two ranks' shards reproduce the single-rank logits
    parts = [pipe.score(pipe.prepare(table.shard(r, 2))).clone() for r in range(2)]
    assert torch.equal(torch.cat(parts), a)


@pytest.mark.parametrize("modalities,layers,N,chunk", [
    (["floor_rgb_texture"], 18, 70, 32),                                   # one surface: 16-byte pixels, a partly filled last chunk
    (["ceiling_rgb_texture", "floor_rgb_texture"], 18, 37, 37),            # two surfaces: 12-byte groups of a 32-byte pixel, the padding channels zeroed
    (["floor_rgb_texture"], 18, 1100, 1100),                               # a launch in costly-first order (>= 1025 renders): tiles follow the render, not the workgroup id
])
def test_tiles_from_the_densify_kernel_equal_the_tile_kernel(modalities, layers, N, chunk):
    """salve_bev_densify_tiles (the densify kernel writes every render's verifier tile in its last phase, from the L2) against the two-launch
    form salve_bev_densify + salve_bev_tile_pairs: the tile buffer, the BEV images and the logits must agree bit for bit -- same integer
    taps, same table, same whole-group stores; only who reads the image differs.  Tile buffers start as NaNs: every sample's pixels, padding
    channels included, must be written by the fused phase exactly where the tile kernel writes them."""
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    dev = torch.device("cuda:0")
    P = 6
    panos = [synthetic.make_pano(i, scene="cluttered" if i % 2 else "box") for i in range(P)]
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=modalities)).eval()
    synthetic.trained_looking_batchnorm(model)
    table = synthetic.make_hypotheses(N, P, seed=3)
    table.t[5] = 40.0           # a render with no point inside the window: an empty image still makes a (normalised-zero) tile in both forms
    outs = []
    for fuse in (True, False):
        pipe = RenderVerifyPipeline(model, dev, chunk=chunk, overlap=False, streams=1, fuse_tiles=fuse)
        pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
        prep = pipe.prepare(table)
        pipe.tile_bufs[0].view(torch.int16).fill_(0x7E00)
        logits = pipe.score(prep).clone()
        torch.cuda.synchronize()
        pipe.check("fused tiles test")
        last = N - (N - 1) // chunk * chunk          # samples of the last chunk: what the buffers hold now
        outs.append((logits, pipe.tile_bufs[0][:last].clone(), pipe.bevs[0][: last * len(pipe.surfaces)].clone(), pipe.valid_mask(prep)))
        del pipe
    (la, ta, ba, va), (lb, tb, bb, vb) = outs
    assert torch.isfinite(ta.float()).all(), "a pixel of the sample was not written by the fused phase"
    assert torch.equal(ta.view(torch.int16), tb.view(torch.int16))
    assert torch.equal(ba, bb) and torch.equal(la, lb) and (va == vb).all() and not va[5]
