"""SURVEY section 8f row 3 on the GPU: the panorama resize kernel against the oracle's restatement of cv2.resize, the
device-resident pano store against the host loader, and disk -> predictions for a floor through the fused pipeline
against the un-fused path (tiles written to disk, read back, verified)."""

import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import bev_oracle as bo  # noqa: E402
from salve_amd import evaluate, ingest, synthetic  # noqa: E402
from salve_amd.common.sim2 import Sim2  # noqa: E402
from salve_amd.utils import image_io  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("src_hw,dst_hw", [((1024, 2048), (512, 1024)), ((500, 1000), (512, 1024)), ((768, 1536), (512, 1024)),
                                           ((64, 128), (32, 64)), ((512, 1024), (512, 1024))])
def test_pano_resize_kernel_matches_oracle(src_hw, dst_hw):
    rng = np.random.default_rng(src_hw[0])
    img = rng.integers(0, 256, size=(2, *src_hw, 3), dtype=np.uint8)
    got = ingest.resize_rgb_on_device(torch.from_numpy(img).to(DEV), dst_hw).cpu().numpy()
    for k in range(2):
        assert np.array_equal(got[k], bo.resize_pano_u8(img[k], dst_hw))


def make_floor(tmp_path, n_panos=4, n_hyp=6, reverse_rooms=False, far=()):
    """A synthetic building on disk in the reference's layout: 2048x1024 JPEG panoramas, 1024x512 .depth.png maps,
    Sim(2) hypothesis files.  reverse_rooms: partial-room numbers run against the pano ids, so the file-name order of
    a pair's tiles disagrees with (i1, i2) wherever i1 < i2.  far: hypotheses whose translation puts pano i1 outside the
    BEV window altogether."""
    raw, depth_root, hyp_root = tmp_path / "zind", tmp_path / "depth", tmp_path / "hyp"
    (raw / "0003" / "panos").mkdir(parents=True)
    fpaths = {}
    for i in range(n_panos):
        rgb, depth = synthetic.make_pano(i)
        big = np.repeat(np.repeat(rgb, 2, axis=0), 2, axis=1)            # 2048 x 1024
        fp = raw / "0003" / "panos" / f"floor_01_partial_room_{(n_panos - 1 - i) if reverse_rooms else i:02d}_pano_{i + 3}.jpg"
        image_io.write_jpeg(str(fp), big)
        image_io.write_depth_png(str(depth_root / "0003" / f"{fp.stem}.depth.png"), depth)
        fpaths[i + 3] = str(fp)
    hyp = synthetic.make_hypotheses(n_hyp, n_panos, seed=2)
    for j in range(n_hyp):
        label = "gt_alignment_approx" if j % 3 == 0 else "incorrect_alignment"
        d = hyp_root / "0003" / "floor_01" / label
        d.mkdir(parents=True, exist_ok=True)
        if j in far:
            hyp.t[j] = np.array([40.0, -35.0], dtype=np.float32)
        Sim2(hyp.R[j].astype(np.float64), hyp.t[j].astype(np.float64), 1.0).save_as_json(
            str(d / f"{int(hyp.i1[j]) + 3}_{int(hyp.i2[j]) + 3}__door_{j}_0_{'identity' if j % 2 else 'rotated'}.json"))
    return raw, depth_root, hyp_root, fpaths


def test_pano_store_equals_host_loader(tmp_path):
    raw, depth_root, _, fpaths = make_floor(tmp_path)
    store = ingest.PanoStore(DEV).load(ingest.floor_pano_fpaths(str(raw), "0003"), str(depth_root), "0003", [3, 4, 5, 6])
    assert len(store) == 4 and store.rgb.shape == (4, 512, 1024, 3) and store.depth.shape == (4, 512, 1024)
    for pid, k in store.index.items():
        rgb = bo.resize_pano_u8(image_io.read_rgb(fpaths[pid]), (512, 1024))
        assert np.array_equal(store.rgb[k].cpu().numpy(), rgb)
        assert np.array_equal(store.depth[k].cpu().numpy().view(np.uint16), image_io.read_depth_png(str(depth_root / "0003" / f"{Path(fpaths[pid]).stem}.depth.png")))
    with pytest.raises((ValueError, FileNotFoundError)):
        ingest.PanoStore(DEV).load({3: fpaths[3]}, str(tmp_path / "nowhere"), "0003", [3])


def test_score_floor_fused_equals_unfused(tmp_path):
    """Same floor, two routes: (a) disk -> device -> render -> verify -> batch files; (b) the reference's two scripts'
    route: render tiles to JPEG, read them back through the dataset, verify.  JPEG is lossy, so (b)'s tiles differ from
    (a)'s by compression noise: predictions are compared loosely, names and labels exactly."""
    from salve_amd.dataset.zind_data import ZindData
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.utils import bev_rendering_utils as bru
    from salve_amd import train_utils
    from tests.test_gpu_dataset import config

    raw, depth_root, hyp_root, fpaths = make_floor(tmp_path)
    torch.manual_seed(3)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    m = ingest.score_floor(model, torch.device(DEV), str(raw), str(depth_root), str(hyp_root), str(tmp_path / "bev"), "0003", "floor_01",
                           str(tmp_path / "preds_fused"), batch_size=4, chunk=4)
    files = sorted((tmp_path / "preds_fused").glob("batch_*.json"))
    assert [f.name for f in files] == ["batch_0.json", "batch_1.json"]
    fused = [json.load(open(f)) for f in files]
    hyps = ingest.load_floor_hypotheses(str(hyp_root), "0003", "floor_01")
    assert sum((g["y_true"] for g in fused), []) == hyps.label.tolist() == [1, 1, 0, 0, 0, 0]
    assert 0 <= m["precision"] <= 1

    # (b) un-fused: write the floor tiles exactly as scripts/render_dataset_bev.py would
    img_fpaths = ingest.floor_pano_fpaths(str(raw), "0003")
    for j in range(len(hyps)):
        label = "gt_alignment_approx" if hyps.label[j] else "incorrect_alignment"
        for surface in ("floor", "ceiling"):
            bru.generate_texture_maps_for_pair(img_fpaths, surface, hyps.fpaths[j], int(hyps.pair_idx[j]), label, str(tmp_path / "bev"), "0003",
                                               "floor_01", str(depth_root), ["rgb_texture"], None, None)
    names = hyps.tile_names(str(tmp_path / "bev"), img_fpaths)
    written = {n for pair in names for n in pair if Path(n).exists()}
    assert sum((g["fp0"] for g in fused), []) == [n[0] for n in names]
    fused_by_name = {}
    for g in fused:
        for k in range(len(g["y_hat"])):
            fused_by_name[g["fp0"][k]] = (g["y_hat"][k], g["y_hat_probs"][k])
    # read back the pairs whose two floor tiles exist (a render with no point in the window writes nothing)
    for a, b in names:
        if a in written and b in written:
            x = train_utils.get_val_test_transform(config("", ["floor_rgb_texture"]))(image_io.read_rgb(a), image_io.read_rgb(b))
            with torch.no_grad():
                probs = torch.softmax(model.cuda()(x[0][None], x[1][None], None, None, None, None), 1)[0]
            y, p = fused_by_name[a]
            assert abs(float(probs[y]) - p) < 0.1


def test_batched_floor_renderer_writes_the_same_files_as_the_pairwise_one(tmp_path):
    """render_dataset.render_building_floor_pairs (one GPU batch per floor) against bev_rendering_utils.
    generate_texture_maps_for_pair called once per (hypothesis, surface), as scripts/render_dataset_bev.py does: same
    file set, same bytes; and a second call writes nothing (skip-if-exists)."""
    from salve_amd import render_dataset
    from salve_amd.utils import bev_rendering_utils as bru

    raw, depth_root, hyp_root, _ = make_floor(tmp_path)
    n = render_dataset.render_pairs(1, str(depth_root), str(tmp_path / "bev_batched"), str(raw), str(hyp_root), None, ["rgb_texture"],
                                    None, "0003", device=DEV)
    hyps = ingest.load_floor_hypotheses(str(hyp_root), "0003", "floor_01")
    img_fpaths = ingest.floor_pano_fpaths(str(raw), "0003")
    for j in range(len(hyps)):
        label = "gt_alignment_approx" if hyps.label[j] else "incorrect_alignment"
        for surface in ("floor", "ceiling"):
            bru.generate_texture_maps_for_pair(img_fpaths, surface, hyps.fpaths[j], int(hyps.pair_idx[j]), label, str(tmp_path / "bev_pairwise"),
                                               "0003", "floor_01", str(depth_root), ["rgb_texture"], None, None)
    a = sorted(p.relative_to(tmp_path / "bev_batched") for p in (tmp_path / "bev_batched").rglob("*.jpg"))
    b = sorted(p.relative_to(tmp_path / "bev_pairwise") for p in (tmp_path / "bev_pairwise").rglob("*.jpg"))
    assert a == b and len(a) == n and n > 0
    for rel in a:
        assert (tmp_path / "bev_batched" / rel).read_bytes() == (tmp_path / "bev_pairwise" / rel).read_bytes()
    assert render_dataset.render_building_floor_pairs(str(depth_root), str(tmp_path / "bev_batched"), str(hyp_root), str(raw), "0003", "floor_01",
                                                      None, ["rgb_texture"], device=DEV) == 0
    with pytest.raises(ValueError):
        render_dataset.render_pairs(1, "", "", "", "", None, ["rgb_texture"], "train", "0003")
    with pytest.raises(NotImplementedError):
        render_dataset.render_building_floor_pairs("", "", "", "", "0003", "floor_01", None, ["layout"])


def test_fused_channel_order_follows_the_sorted_tile_names(tmp_path):
    """The reference's dataset hands the verifier a pair's two tiles in FILE-NAME order (zind_data.py:110), which follows
    the pano stems, not (i1, i2).  Here the partial-room numbers run against the pano ids, so for every hypothesis with
    i1 < i2 the posed render is the SECOND image.  The fused path must feed exactly what the un-fused route feeds:
    (a) lossless: tiles rendered through the facade, ordered by the dataset's own grouping rule
        (zind_data.get_tuples_from_fpath_list on the names) -> same logits as the fused path, to the last bit;
    (b) through JPEG files on disk and the dataset rule: logits agree within compression noise in the matched order, and
        the other order is further away."""
    from salve_amd import train_utils
    from salve_amd.dataset import zind_data
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline
    from salve_amd.utils import bev_rendering_utils as bru
    from tests.test_gpu_dataset import config

    raw, depth_root, hyp_root, fpaths = make_floor(tmp_path, reverse_rooms=True)
    torch.manual_seed(3)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    dev = torch.device(DEV)
    hyps = ingest.load_floor_hypotheses(str(hyp_root), "0003", "floor_01")
    img_fpaths = ingest.floor_pano_fpaths(str(raw), "0003")
    swap = hyps.swap(img_fpaths)
    assert swap.tolist() == (hyps.i1 < hyps.i2).tolist() and swap.any() and not swap.all()
    store = ingest.PanoStore(dev).load(img_fpaths, str(depth_root), "0003", np.concatenate([hyps.i1, hyps.i2]))
    pipe = RenderVerifyPipeline(model, dev, chunk=4)
    pipe.set_panos(store.rgb, store.depth)
    prepared = pipe.prepare(hyps.table(store, img_fpaths))
    fused = pipe.score(prepared).cpu()
    assert pipe.valid_mask(prepared).all()
    pipe.check()

    args = config(str(tmp_path / "bev"), ["floor_rgb_texture"])
    tf = train_utils.get_val_test_transform(args)
    for j in range(len(hyps)):
        label = "gt_alignment_approx" if hyps.label[j] else "incorrect_alignment"
        for surface in ("floor", "ceiling"):
            bru.generate_texture_maps_for_pair(img_fpaths, surface, hyps.fpaths[j], int(hyps.pair_idx[j]), label, str(tmp_path / "bev"), "0003",
                                               "floor_01", str(depth_root), ["rgb_texture"], None, None)
    names = hyps.tile_names(str(tmp_path / "bev"), img_fpaths)
    worse = 0
    for j in range(len(hyps)):
        label_dir = tmp_path / "bev" / ("gt_alignment_approx" if hyps.label[j] else "incorrect_alignment") / "0003"
        mine = [str(p) for p in label_dir.glob(f"pair_{int(hyps.pair_idx[j])}___*.jpg")]
        (f1, f2, y), = zind_data.get_tuples_from_fpath_list(mine, int(hyps.label[j]), args)   # the dataset's ordering rule
        assert (f1, f2) == names[j] and y == hyps.label[j]
        # (a) lossless tiles in that order
        a = SimpleNamespace(img_i1=img_fpaths[int(hyps.i1[j])], img_i2=img_fpaths[int(hyps.i2[j])],
                            depth_i1=str(depth_root / "0003" / f"{Path(img_fpaths[int(hyps.i1[j])]).stem}.depth.png"),
                            depth_i2=str(depth_root / "0003" / f"{Path(img_fpaths[int(hyps.i2[j])]).stem}.depth.png"),
                            scale=0.001, crop_ratio=80 / 512, crop_z_range=[-float("inf"), -1.0])
        S = Sim2.from_json(hyps.fpaths[j])
        img1, img2 = bru.render_bev_pair(a, "0003", "floor_01", int(hyps.i1[j]), int(hyps.i2[j]), S, False)
        first, second = (img2, img1) if swap[j] else (img1, img2)
        x = tf(first, second)
        with torch.no_grad():
            lossless = model.cuda()(x[0][None], x[1][None], None, None, None, None).cpu()[0]
        assert torch.equal(lossless, fused[j]), (j, lossless, fused[j])
        # (b) the JPEG route
        xj = tf(image_io.read_rgb(f1), image_io.read_rgb(f2))
        with torch.no_grad():
            matched = model(xj[0][None], xj[1][None], None, None, None, None).cpu()[0]
            crossed = model(xj[1][None], xj[0][None], None, None, None, None).cpu()[0]
        e_m, e_c = float((matched - fused[j]).abs().max()), float((crossed - fused[j]).abs().max())
        assert e_m < 0.05 * max(1.0, float(fused[j].abs().max())), (j, e_m)
        worse += e_c > e_m
    assert worse >= len(hyps) - 1    # the crossed order is (almost always) further from the fused logits than the matched one


def test_hypotheses_without_points_in_the_window_are_dropped(tmp_path):
    """render_bev_image returns None when no point falls inside the BEV window (bev_rendering_utils.py:279-280), the pair
    then writes no tile (:623-627) and never reaches scripts/test.py.  The fused route reports those hypotheses through the
    in-window counts and writes no prediction for them."""
    from salve_amd.models.early_fusion import EarlyFusionCEResnet

    raw, depth_root, hyp_root, _ = make_floor(tmp_path, n_hyp=6, far=(1, 4))
    torch.manual_seed(3)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    m = ingest.score_floor(model, torch.device(DEV), str(raw), str(depth_root), str(hyp_root), str(tmp_path / "bev"), "0003", "floor_01",
                           str(tmp_path / "preds"), batch_size=4, chunk=4)
    assert m["num_hypotheses"] == 6 and m["num_dropped_no_points_in_window"] == 2
    got = [json.load(open(f)) for f in sorted((tmp_path / "preds").glob("batch_*.json"))]
    hyps = ingest.load_floor_hypotheses(str(hyp_root), "0003", "floor_01")
    far_names = {Path(p).stem.split("__")[-1] for p in hyps.fpaths if "_1_0_" in Path(p).name or "_4_0_" in Path(p).name}
    assert len(far_names) == 2
    kept = sum((g["fp0"] for g in got), [])
    assert len(kept) == 4 and not any(u in n for n in kept for u in far_names)


def _synthetic_pose_graph(pano_ids, seed=5):
    """A floor's rooms and W/D/O objects, in the attributes layout.layout_pair_specs reads (salve/common/posegraph2d.py nodes,
    salve/common/wdo.py:47-50)."""
    rng = np.random.default_rng(seed)
    wdo = lambda t, a, b: SimpleNamespace(type=t, vertices_local_2d=np.array([a, b], dtype=np.float64))
    nodes = {}
    for pid in pano_ids:
        n = int(rng.integers(4, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(0.8, 2.6, n)
        room = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
        objs = {"doors": [], "windows": [], "openings": []}
        for j in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, n))
            p, q = room[a], room[(a + 1) % n]
            t0, t1 = np.sort(rng.uniform(0, 1, 2))
            kind = ("doors", "windows", "openings")[j % 3]
            objs[kind].append(wdo(kind, p + t0 * (q - p), p + t1 * (q - p)))
        nodes[int(pid)] = SimpleNamespace(room_vertices_local_2d=room, **objs)
    return SimpleNamespace(nodes=nodes)


@pytest.mark.parametrize("modalities", [["layout"], ["ceiling_rgb_texture", "floor_rgb_texture", "layout"]])
def test_fused_layout_modalities_equal_the_unfused_route(tmp_path, modalities):
    """The rasterised-layout modality through the FUSED pipeline (early_fusion.py:24-32, 59-60; rasteriser
    bev_rendering_utils.py:48-156): per hypothesis, salve_layout_rasterise of pano i1's posed layout -> salve_bev_tile_pairs into
    the channels behind the texture maps' -> the 6- / 18-channel stem.  Against the un-fused route -- the facade's
    rasterize_room_layout_pair / render_bev_pair images, the reference's val transform, `model(x1 .. x6)` in the dataset's order
    (zind_data.py:26, 110: ceiling 1, 2, floor 1, 2, layout 1, 2, each pair in FILE-NAME order) -- the logits must agree to the
    last bit.  The partial-room numbers run against the pano ids, so the file-name order differs from (i1, i2) for some pairs."""
    from salve_amd import layout, train_utils
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline
    from salve_amd.utils import bev_rendering_utils as bru
    from tests.test_gpu_dataset import config

    raw, depth_root, hyp_root, fpaths = make_floor(tmp_path, reverse_rooms=True)
    torch.manual_seed(4)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=modalities)).eval()
    dev = torch.device(DEV)
    hyps = ingest.load_floor_hypotheses(str(hyp_root), "0003", "floor_01")
    img_fpaths = ingest.floor_pano_fpaths(str(raw), "0003")
    swap = hyps.swap(img_fpaths)
    assert swap.any() and not swap.all()
    store = ingest.PanoStore(dev).load(img_fpaths, str(depth_root), "0003", np.concatenate([hyps.i1, hyps.i2]))
    pano_ids = [pid for pid, _ in sorted(store.index.items(), key=lambda kv: kv[1])]
    graph = _synthetic_pose_graph(pano_ids)
    table = hyps.table(store, img_fpaths)
    pipe = RenderVerifyPipeline(model, dev, chunk=4)     # 6 hypotheses: two chunks, the second partly filled
    pipe.set_panos(store.rgb, store.depth)
    with pytest.raises(RuntimeError, match="layout"):
        pipe.prepare(table)
    prepared = pipe.prepare(table, layouts=layout.FusedLayouts.from_pose_graph(table, pano_ids, graph, hyps.s))
    fused = pipe.score(prepared).cpu()
    assert pipe.valid_mask(prepared).all()
    pipe.check("fused layout")

    tf = train_utils.get_val_test_transform(config(str(tmp_path / "bev"), modalities))
    for j in range(len(hyps)):
        i1, i2 = int(hyps.i1[j]), int(hyps.i2[j])
        S = Sim2.from_json(hyps.fpaths[j])
        order = (lambda a, b: (b, a)) if swap[j] else (lambda a, b: (a, b))
        imgs = []
        if "floor_rgb_texture" in modalities:
            for surface, zr in (("ceiling", [0.5, float("inf")]), ("floor", [-float("inf"), -1.0])):
                a = SimpleNamespace(img_i1=img_fpaths[i1], img_i2=img_fpaths[i2],
                                    depth_i1=str(depth_root / "0003" / f"{Path(img_fpaths[i1]).stem}.depth.png"),
                                    depth_i2=str(depth_root / "0003" / f"{Path(img_fpaths[i2]).stem}.depth.png"),
                                    scale=0.001, crop_ratio=80 / 512, crop_z_range=zr)
                imgs += order(*bru.render_bev_pair(a, "0003", "floor_01", i1, i2, S, False))
        imgs += order(*bru.rasterize_room_layout_pair(S, graph, "0003", "floor_01", i1, i2))
        x = [t[None] for t in tf(*imgs)] + [None] * (6 - len(imgs))
        with torch.no_grad():
            unfused = model.cuda()(*x).cpu()[0]
        assert torch.equal(unfused, fused[j]), (modalities, j, unfused, fused[j])
