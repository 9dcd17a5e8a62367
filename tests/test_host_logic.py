"""Host-side logic of salve_amd (no GPU): tables, packing, program builder, sharding, API mirrors."""

import os
import json
import re
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import bev_oracle as bo
from oracle import resnet_oracle as ro
from salve_amd import _lib, synthetic
from salve_amd.common.bevparams import BEVParams, get_line_width_by_resolution
from salve_amd.common.sim2 import Sim2
from salve_amd.models import hip_resnet
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.rasteriser import linear_resize_taps, normalisation_lut, pack_hypotheses
from salve_amd.utils import image_io
from salve_amd.utils.hohonet_pano_utils import get_sphere_factors, get_uni_sphere_xyz
from salve_amd.utils.rotation_utils import rotmat2d

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "salve_hip.h").read_text()
    declared = set(re.findall(r"\b(salve_[a-z0-9_]+)\s*\(", header))
    declared -= {"salve_status_t"}
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.salve_hip_version() == _lib.EXPECTED_ABI == int(re.search(r"#define SALVE_HIP_ABI_VERSION (\d+)", header).group(1))
    assert lib.salve_last_error() is not None


def test_product_library_reads_no_environment_variable():
    """Kernel selection is an argument (salve_resnet_create flags, ABI 5), never the environment: a stray variable in a user's
    shell must not change which kernels a production run uses.  The built library does not even import getenv."""
    import subprocess

    _lib.load()
    undefined = subprocess.run(["nm", "-D", "--undefined-only", str(_lib.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    assert "getenv" not in undefined
    for src in sorted((ROOT / "salve_amd" / "csrc").glob("*")):
        text = src.read_text()
        for i, line in enumerate(text.splitlines()):
            if "getenv(" in line:   # only inside the ablation build's #ifdef blocks
                before = text.splitlines()[max(0, i - 3):i + 1]
                assert any("SALVE_BUILD_ABLATIONS" in b for b in before), f"{src.name}:{i + 1}: {line.strip()}"


def test_product_has_no_cpu_path():
    from salve_amd.rasteriser import BevRasteriser

    with pytest.raises(_lib.SalveHipError):
        BevRasteriser(torch.device("cpu"))
    with pytest.raises(_lib.SalveHipError):
        hip_resnet.HipResNet({}, 50, torch.device("cpu"))
    if not torch.cuda.is_available():
        from salve_amd.utils import bev_rendering_utils

        with pytest.raises(_lib.SalveHipError):
            bev_rendering_utils.render_bev_image(BEVParams(), np.zeros((4, 6)), False)


def test_product_never_imports_the_oracle():
    for p in (ROOT / "salve_amd").rglob("*.py"):
        text = re.sub(r"#.*", "", p.read_text())
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), p


def test_sphere_table_matches_oracle():
    for hw in ((512, 1024), (64, 128)):
        assert np.array_equal(get_uni_sphere_xyz(*hw), bo.sphere_table(*hw))
        r, z, c, s = get_sphere_factors(*hw)
        t = bo.sphere_table(*hw)
        assert np.array_equal(r[:, None] * c[None, :], t[..., 0]) and np.array_equal(z, t[:, 0, 2])


def test_resize_taps_and_lut_match_oracle():
    for dst, src in ((234, 501), (224, 501), (512, 1000), (300, 299)):
        s0, s1, a0, a1 = bo._linear_coeffs(dst, src)
        t = linear_resize_taps(dst, src)
        assert np.array_equal(t, np.stack([s0, s1, a0, a1], -1))
    lut = normalisation_lut()
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    t = bo.tile_from_bev(img, resize_hw=(16, 16), crop_hw=(16, 16))
    for c in range(3):
        assert np.array_equal(t[c].reshape(-1), lut[c])


def test_host_resize_matches_oracle_and_box_average():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(40, 60, 3), dtype=np.uint8)
    assert np.array_equal(image_io.resize_linear_u8(img, (17, 23)), bo.resize_linear_u8(img, (17, 23)))
    half = image_io.resize_linear_u8(img, (20, 30))
    a = img.astype(int)
    assert np.array_equal(half, (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2)
    assert image_io.resize_linear_u8(img, (40, 60)) is img


def test_hypothesis_packing_and_sharding():
    hyp = synthetic.make_hypotheses(100, 7, seed=3)
    assert hyp.R.dtype == np.float32 and hyp.t.dtype == np.float32 and (hyp.i1 != hyp.i2).all()
    h = pack_hypotheses(hyp.i1, np.zeros(100), hyp.R, hyp.t, np.ones(100))
    raw = h.view(np.uint8).reshape(100, 40)
    assert np.array_equal(raw[:, 0:4].copy().view(np.int32)[:, 0], hyp.i1)
    assert np.array_equal(raw[:, 8:24].copy().view(np.float32), hyp.R.reshape(100, 4))
    assert np.array_equal(raw[:, 24:32].copy().view(np.float32), hyp.t)
    parts = [hyp.shard(r, 8) for r in range(8)]
    assert sum(len(p) for p in parts) == 100
    assert np.array_equal(np.concatenate([p.i1 for p in parts]), hyp.i1)
    assert np.array_equal(synthetic.make_hypotheses(100, 7, seed=3).t, hyp.t)  # deterministic


def test_sim2_mirror():
    g = np.load(ROOT / "tests" / "golden" / "g5_sim2.npz")
    S = Sim2(R=rotmat2d(33.3), t=np.array([0.25, -1.75]), s=1.0)
    assert S.rotation.dtype == np.float32 and np.array_equal(S.rotation, g["R32"]) and np.array_equal(S.translation, g["t32"])
    assert np.array_equal(S.transform_from(g["pts"]), g["S_pts"])
    assert np.array_equal(BEVParams().bevimg_Sim2_world.transform_from(g["pts"]), g["img_pts"])
    # algebra (reference tests/common/test_sim2.py)
    a = Sim2(rotmat2d(90), np.array([1.0, 2.0]), 3.0)
    I = a.compose(a.inverse())
    assert np.allclose(I.rotation, np.eye(2), atol=1e-6) and np.allclose(I.translation, 0, atol=1e-6) and np.isclose(I.scale, 1)
    assert a == Sim2.from_matrix(a.matrix)
    pts = np.array([[1.0, 0.0], [0.0, 2.0]])
    assert np.allclose(a.inverse().transform_from(a.transform_from(pts)), pts, atol=1e-5)
    with pytest.raises(ValueError):
        Sim2(np.eye(3), np.zeros(2), 1.0)
    with pytest.raises(ZeroDivisionError):
        Sim2(np.eye(2), np.zeros(2), 0.0)
    with pytest.raises(ValueError):
        a.transform_from(np.zeros((3, 3)))


def test_sim2_json_roundtrip(tmp_path):
    a = Sim2(rotmat2d(-12.5), np.array([0.5, -2.0]), 1.0)
    f = tmp_path / "d" / "0_1__door_0_0_identity.json"
    a.save_as_json(f)
    d = json.loads(f.read_text())
    assert set(d) == {"R", "t", "s"} and len(d["R"]) == 4 and len(d["t"]) == 2
    b = Sim2.from_json(f)
    assert np.array_equal(a.rotation, b.rotation) and np.array_equal(a.translation, b.translation) and b.scale == 1.0


def test_bevparams_mirror():
    p = BEVParams(img_h=20, img_w=20, meters_per_px=0.5)
    assert p.xlims == [-5, 5] and p.ylims == [-5, 5]
    got = p.bevimg_Sim2_world.transform_from(np.array([[2, 2], [-5, -5], [5, 5]]))
    assert np.allclose(got, [[14, 14], [0, 0], [20, 20]])
    assert [get_line_width_by_resolution(r) for r in (0.005, 0.01, 0.02)] == [30, 15, 8]


def test_model_state_dict_layout_and_program():
    torch.manual_seed(0)
    for layers, modalities, cin in ((50, ["floor_rgb_texture"], 6), (18, ["layout"], 6),
                                    (152, ["ceiling_rgb_texture", "floor_rgb_texture"], 12)):
        m = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=modalities))
        assert list(m.conv1.weight.shape) == [64, cin, 7, 7]
        assert set(m.state_dict()) == set(ro.expected_state_dict_keys(layers, cin // 3))
    m = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"]))
    sd = {"module." + k: v for k, v in m.state_dict().items()}  # DataParallel checkpoints
    ops, w, p, k, cin_p = hip_resnet.build_program(sd, 50)
    # stem conv + max-pool + 16 blocks x 3 convolutions + avg-pool/fc; the 4 projection shortcuts ride along in their
    # block's last convolution as a second source (include/salve_hip.h: in2_buf)
    assert cin_p == 8 and len(ops) == 1 + 1 + 16 * 3 + 1
    convs = ops[ops["op"] == 0]
    assert (convs["in2_buf"] != hip_resnet.NO_BUF).sum() == 4
    assert (convs["Cout"] % 64 == 0).all() and ((convs["KH"] * convs["KW"] * convs["Cin"]) % 64 == 0).all()
    assert convs["w_off"][-1] + convs["Cout"][-1] * (convs["KH"][-1] * convs["KW"][-1] * convs["Cin"][-1] + convs["Cin2"][-1]) == w.size
    flops = 2 * (convs["Ho"].astype(np.int64) * convs["Wo"] * convs["Cout"] * (convs["KH"] * convs["KW"] * convs["Cin"] + convs["Cin2"])).sum()
    assert 8.3e9 < flops < 9.6e9  # 8.41 GFLOP algorithmic + the stem's zero padding
    assert tuple(ops[-1][["Hi", "Wi", "Cin", "Cout"]]) == (7, 7, 2048, 2)


def test_bn_folding_is_exact_algebra():
    torch.manual_seed(1)
    w = torch.randn(8, 4, 3, 3)
    bn = {"weight": torch.rand(8) + 0.5, "bias": torch.randn(8), "running_mean": torch.randn(8), "running_var": torch.rand(8) + 0.5}
    x = torch.randn(2, 4, 9, 9)
    wf, bf = hip_resnet.fold_bn(w, bn)
    ref = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x, w, None, 1, 1), bn["running_mean"], bn["running_var"],
                                         bn["weight"], bn["bias"], False, 0.0, 1e-5)
    assert torch.allclose(torch.nn.functional.conv2d(x, wf, bf, 1, 1), ref, atol=1e-5)


def test_unsupported_modalities_raise_like_the_reference():
    with pytest.raises(RuntimeError):
        EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture", "layout"]))


def test_training_config_loader(tmp_path):
    from salve_amd.training_config import load_training_config

    y = tmp_path / "c.yaml"
    y.write_text("TrainingConfig:\n    _target_: salve.training_config.TrainingConfig\n    lr_annealing_strategy: poly\n    base_lr: 0.001\n"
                 "    weight_decay: 0.0001\n    num_ce_classes: 2\n    print_every: 10\n    poly_lr_power: 0.9\n    optimizer_algo: adam\n"
                 "    num_layers: 152\n    pretrained: True\n    dataparallel: True\n    resize_h: 234\n    resize_w: 234\n    train_h: 224\n"
                 "    train_w: 224\n    apply_photometric_augmentation: False\n    modalities: [\"ceiling_rgb_texture\", \"floor_rgb_texture\"]\n"
                 "    cfg_stem:\n    num_epochs: 50\n    workers: 15\n    batch_size: 256\n    data_root: /x\n    layout_data_root:\n"
                 "    model_save_dirpath: /y\n    gpu_ids:\n")
    c = load_training_config(str(y))
    assert c.num_layers == 152 and c.resize_h == 234 and c.train_h == 224 and c.modalities == ("ceiling_rgb_texture", "floor_rgb_texture")


def test_checkpoint_loader_accepts_dataparallel_prefix(tmp_path):
    from salve_amd import train_utils

    args = SimpleNamespace(modalities=["floor_rgb_texture"], num_layers=18, pretrained=False, num_ce_classes=2, dataparallel=True)
    torch.manual_seed(0)
    src = EarlyFusionCEResnet(18, False, 2, args)
    ck = tmp_path / "train_ckpt.pth"
    torch.save({"epoch": 3, "state_dict": {"module." + k: v for k, v in src.state_dict().items()}}, ck)
    dst = EarlyFusionCEResnet(18, False, 2, args)
    train_utils.load_model_checkpoint(str(ck), dst, args)
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    with pytest.raises(RuntimeError):
        train_utils.load_model_checkpoint(str(tmp_path / "missing.pth"), dst, args)


def _run_py(code: str, extra_path=None):
    import subprocess
    import sys as _sys

    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([str(ROOT)] + ([str(extra_path)] if extra_path else []))
    r = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_install_as_salve_alias_when_no_salve_package_exists():
    """No `salve` importable: the whole package is aliased (hot path only).  Run in a fresh interpreter."""
    out = _run_py(
        "import salve_amd; salve_amd.install_as_salve()\n"
        "import salve.common.sim2 as s2, salve.utils.bev_rendering_utils as b\n"
        "from salve_amd.common.sim2 import Sim2\n"
        "assert s2.Sim2 is Sim2 and callable(b.generate_texture_maps_for_pair)\n"
        "print(b.bev_fname_from_img_fpath(58, 'opening_0_0_rotated', 'floor', '/x/floor_01_partial_room_01_pano_13.jpg'))\n")
    assert out.strip() == "pair_58___opening_0_0_rotated_floor_rgb_floor_01_partial_room_01_pano_13.jpg"


def test_install_as_salve_overlays_a_real_package(tmp_path):
    """A `salve` package is importable (here: a throw-away fake with the modules the reference's drivers import next to
    the hot path -- scripts/test.py:17-23, scripts/render_dataset_bev.py:21-26): only the hot-path submodules are
    replaced, everything else keeps resolving to the package's own files."""
    pkg = tmp_path / "salve"
    for d in ("", "utils", "common", "dataset", "models"):
        (pkg / d).mkdir(parents=True, exist_ok=True)
        (pkg / d / "__init__.py").write_text("")
    (pkg / "utils" / "io.py").write_text("MARK = 'real io'\n")
    (pkg / "utils" / "avg_meter.py").write_text("MARK = 'real avg_meter'\n")
    (pkg / "utils" / "logger_utils.py").write_text("MARK = 'real logger'\n")
    (pkg / "common" / "posegraph2d.py").write_text("MARK = 'real posegraph2d'\n")
    (pkg / "dataset" / "hnet_prediction_loader.py").write_text("MARK = 'real loader'\n")
    (pkg / "utils" / "bev_rendering_utils.py").write_text("MARK = 'the CPU renderer that must be replaced'\n")
    (pkg / "models" / "early_fusion.py").write_text("MARK = 'the torchvision model that must be replaced'\n")
    _run_py(
        "import salve_amd; salve_amd.install_as_salve()\n"
        "import salve, salve.utils.io, salve.utils.avg_meter, salve.utils.logger_utils\n"
        "import salve.common.posegraph2d, salve.dataset.hnet_prediction_loader\n"
        "assert salve is not salve_amd and salve.utils.io.MARK == 'real io' and salve.common.posegraph2d.MARK == 'real posegraph2d'\n"
        "assert salve.dataset.hnet_prediction_loader.MARK == 'real loader' and salve.utils.avg_meter.MARK == 'real avg_meter'\n"
        "import salve.utils.bev_rendering_utils as b, salve.models.early_fusion as ef, salve.common.sim2 as s2\n"
        "from salve.utils import bev_rendering_utils as b2\n"
        "import salve_amd.utils.bev_rendering_utils as mine, salve_amd.models.early_fusion as mine_ef, salve_amd.common.sim2 as mine_s2\n"
        "assert b is mine and b2 is mine and ef is mine_ef and s2 is mine_s2 and not hasattr(b, 'MARK')\n"
        "assert salve.utils.bev_rendering_utils is mine\n", extra_path=tmp_path)


def test_cluttered_scene_is_a_box_room_with_occluders():
    """synthetic.make_pano(scene="cluttered"): same walls as the box room of that index; furniture only ever shortens a ray,
    the door opening only lengthens it; all returns stay inside the z-slice range of the reference's z-order ([-2, 2))."""
    for idx in (0, 3):
        box = synthetic.make_box_room_depth_mm(idx, 128, 256).astype(np.int64)
        clut = synthetic.make_cluttered_room_depth_mm(idx, 128, 256).astype(np.int64)
        assert clut.shape == box.shape and clut.min() > 300
        shorter, longer = (clut < box).mean(), (clut > box).mean()
        assert 0.02 < shorter < 0.6 and 0.0005 < longer < 0.1, (shorter, longer)
        z = clut / 1000.0 * get_uni_sphere_xyz(128, 256)[..., 2]
        assert z.min() >= synthetic.FLOOR_Z - 1e-3 and z.max() <= synthetic.CEILING_Z + 1e-3
    assert np.array_equal(synthetic.make_pano(2, 64, 128)[1], synthetic.make_box_room_depth_mm(2, 64, 128))
    assert np.array_equal(synthetic.make_pano(2, 64, 128, scene="cluttered")[0], synthetic.make_pano(2, 64, 128)[0])


def test_bindings_refuse_a_library_of_another_abi(monkeypatch):
    """_lib.load() compares salve_hip_version() with the ABI the ctypes signatures were written for: a stale git-ignored .so or a
    SALVE_HIP_LIB override built from another revision would otherwise be called with shifted arguments."""
    from salve_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "EXPECTED_ABI", _lib.EXPECTED_ABI + 1)
    with pytest.raises(_lib.SalveHipError, match="ABI version"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load().salve_hip_version() == _lib.EXPECTED_ABI


def test_the_product_library_exports_no_development_entry_points():
    """salve_debug_burn lives in the TEST helper library (tests/native), not in libsalve_hip.so."""
    import ctypes

    from salve_amd import _lib

    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    assert not hasattr(lib, "salve_debug_burn")
    helper = ROOT / "tests" / "native" / "libsalve_testhelp.so"
    assert helper.exists() and hasattr(ctypes.CDLL(str(helper)), "salve_debug_burn")


def test_non_finite_weights_are_refused_when_the_program_is_packed():
    """The HIP kernels' ReLUs swallow NaNs (resnet.hip: track4), torch propagates them: a diverged checkpoint must not come out
    as finite logits -- hip_resnet.build_program raises instead."""
    from salve_amd.models import hip_resnet

    torch.manual_seed(0)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["layout"])).eval()
    sd = model.state_dict()
    hip_resnet.build_program(sd, 18)
    sd["resnet.layer3.0.conv1.weight"][3, 2, 1, 1] = float("nan")
    with pytest.raises(ValueError, match="non-finite"):
        hip_resnet.build_program(sd, 18)


def test_render_order_sorts_by_panorama_inside_chunks_only():
    """pipeline.render_order: renders are issued in panorama order INSIDE a chunk (L2 locality of the splat kernel); the tile jobs
    find a hypothesis's render by its rank, so nothing a caller sees changes order."""
    from salve_amd.pipeline import render_order

    rng = np.random.default_rng(0)
    for n, chunk in ((0, 8), (5, 8), (64, 16), (100, 32), (4096, 4096)):
        i1 = rng.integers(0, 7, n)
        order, rank = render_order(i1, chunk)
        assert sorted(order.tolist()) == list(range(n)) and np.array_equal(rank[order], np.arange(n))
        for lo in range(0, n, chunk):
            blk = order[lo:lo + chunk]
            assert blk.min() >= lo and blk.max() < lo + chunk          # a chunk keeps its hypotheses
            assert (np.diff(i1[blk]) >= 0).all()                         # ... in panorama order
            for p in np.unique(i1[blk]):                                 # ... stable: ties keep the table's order
                assert (np.diff(blk[i1[blk] == p]) > 0).all()


def test_noisy_scene_is_the_cluttered_scene_with_network_like_depth_errors():
    """synthetic.make_pano(scene="noisy"): the cluttered scene's depth under a smooth bias field (a few %), per-pixel noise (sigma
    0.6 %) and smeared discontinuities -- deterministic per index, same colours, errors of the size a depth network makes."""
    clut = synthetic.make_cluttered_room_depth_mm(3, 128, 256).astype(np.float64)
    noisy = synthetic.make_noisy_room_depth_mm(3, 128, 256).astype(np.float64)
    assert np.array_equal(noisy, synthetic.make_noisy_room_depth_mm(3, 128, 256))
    rel = (noisy - clut) / clut
    assert 0.004 < np.std(rel) < 0.03 and np.abs(np.median(rel)) < 0.02
    assert (np.abs(rel) < 0.05).mean() > 0.98          # the smeared discontinuities are the rest
    assert np.array_equal(synthetic.make_pano(3, 64, 128, scene="noisy")[0], synthetic.make_pano(3, 64, 128)[0])


def test_fused_layout_host_side():
    """The host side of the fused layout modality (no GPU): the modality sets the pipeline accepts (early_fusion.py:24-32), the posed /
    identity layout specs FusedLayouts builds from a pose graph -- pano i1's room and W/D/Os under i2Ti1 (bev_rendering_utils.py:82, 90),
    every panorama's own layout closed like the reference closes it (:76-77), an empty image for a panorama the graph lacks -- and the
    refusal to rasterise without a HIP device."""
    from types import SimpleNamespace

    from salve_amd import _lib, layout, pipeline
    from salve_amd.common.sim2 import Sim2

    assert pipeline.surfaces_for(["layout"]) == [] and pipeline.surfaces_for(["layout", "floor_rgb_texture", "ceiling_rgb_texture"]) == ["ceiling", "floor"]
    with pytest.raises(RuntimeError):
        pipeline.surfaces_for(["layout", "floor_rgb_texture"])
    wdo = lambda t, a, b: SimpleNamespace(type=t, vertices_local_2d=np.array([a, b], dtype=np.float64))
    node = lambda v, d, w, o: SimpleNamespace(room_vertices_local_2d=np.array(v, dtype=np.float64), doors=d, windows=w, openings=o)
    graph = SimpleNamespace(nodes={4: node([[-1, -1], [1, -1], [1, 1], [-1, 1]], [wdo("doors", [1, -0.5], [1, 0.3])], [], []),
                                   7: node([[-2, -1], [1, -1], [1, 1]], [], [wdo("windows", [0, -1], [0.5, -1])], [wdo("openings", [1, 0], [1, 0.5])])})
    table = synthetic.make_hypotheses(5, 3, seed=1)
    table.i1[:] = [0, 1, 0, 1, 0]
    table.i2[:] = [1, 0, 1, 0, 1]
    pano_ids = [4, 7, 9]                         # store index -> pano id; pano 9 is not in the graph (no hypothesis names it)
    fl = layout.FusedLayouts.from_pose_graph(table, pano_ids, graph)
    assert len(fl.posed) == 5 and len(fl.identity) == 3
    for j in range(5):
        S = Sim2(table.R[j], table.t[j], 1.0)
        n1 = graph.nodes[pano_ids[int(table.i1[j])]]
        room, wdos = fl.posed[j]
        closed = np.vstack([n1.room_vertices_local_2d, n1.room_vertices_local_2d[:1]])
        assert np.array_equal(room, S.transform_from(closed))
        objs = list(n1.doors) + list(n1.windows) + list(n1.openings)
        assert [w[0] for w in wdos] == [o.type for o in objs]
        assert all(np.array_equal(w[1], S.transform_from(o.vertices_local_2d)) for w, o in zip(wdos, objs))
    room7, wdos7 = fl.identity[1]
    assert room7.shape == (4, 2) and np.array_equal(room7[0], room7[-1]) and [w[0] for w in wdos7] == ["windows", "openings"]
    assert fl.identity[2][0].shape == (0, 2) and fl.identity[2][1] == []
    with pytest.raises(_lib.SalveHipError):
        layout.pack_layouts(fl.posed, "cpu")


def test_pick_launch_prefers_the_whole_shard_and_cuts_into_equal_whole_rounds():
    """pipeline.pick_launch (pure host arithmetic): the whole shard in one launch when it fits the budget; otherwise the FEWEST equal
    launches, in whole rounds of the 512 resident densify workgroups, never beyond the budget."""
    from salve_amd.pipeline import pick_launch

    MB = 1 << 20
    assert pick_launch(4096, 14 * MB, 140 * 1024 * MB) == 4096                    # config 3: 57 GB of 140
    assert pick_launch(4096, 20 * MB, 140 * 1024 * MB, surfaces=2) == 4096        # config 5: 80 GB of 140
    assert pick_launch(32768, 14 * MB, 140 * 1024 * MB) == 8192                   # fit 10240 -> four equal launches of 8192 rows = 16 rounds each
    n = pick_launch(5000, 14 * MB, 60 * 1024 * MB)                                # fit 4388 -> two launches of 2500 -> 2560 (5 rounds)
    assert n == 2560 and n % 512 == 0
    n = pick_launch(5000, 20 * MB, 60 * 1024 * MB, surfaces=2)                    # fit 3072 -> two launches of 2500 -> 2560 hypotheses = 5120 renders
    assert n == 2560 and (2 * n) % 512 == 0
    assert pick_launch(3, 14 * MB, 20 * MB) == 1                                  # one fits: one at a time
    assert pick_launch(700, 14 * MB, 600 * 14 * MB) == 512                        # rounding UP (to 512) would leave the budget (600): stays inside
    with pytest.raises(RuntimeError):
        pick_launch(8, 14 * MB, MB)
    with pytest.raises(ValueError):
        pick_launch(0, 14 * MB, 1 << 40)
    # a shard with more than 1024 renders is not cut into launches of 1024 or fewer while memory allows more
    assert pick_launch(4096, 14 * MB, 3000 * 14 * MB) * 1 > 1024


def test_every_recorded_experiment_patch_applies_to_its_base_commit(tmp_path):
    """tools/probe/ablations/*.patch are records of measured experiments, pinned to the commit they were made against (MANIFEST.json): each
    must dry-run-apply to that commit's tree (after the patches it builds on), and every patch in the directory must be in the manifest --
    so none of them can rot silently.  Needs the git history (this container; skipped where the tree travels without .git)."""
    import json
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    abl = root / "tools" / "probe" / "ablations"
    man = json.loads((abl / "MANIFEST.json").read_text())
    assert sorted(p.name for p in abl.glob("*.patch")) == sorted(man["patches"]), "a patch without a manifest entry (or the reverse)"
    if not (root / ".git").exists():
        pytest.skip("no git history here")
    bases = {}

    def tree_of(commit):
        if commit not in bases:
            d = tmp_path / ("base_" + commit)
            d.mkdir()
            tar = subprocess.run(["git", "-C", str(root), "archive", commit], capture_output=True, check=True).stdout
            subprocess.run(["tar", "-x", "-C", str(d)], input=tar, check=True)
            bases[commit] = d
        return bases[commit]

    for name, e in man["patches"].items():
        work = tmp_path / ("w_" + name)
        subprocess.run(["cp", "-r", str(tree_of(e.get("base", man["base_commit"]))), str(work)], check=True)
        for pre in e["after"]:
            subprocess.run(["patch", "-p1", "-s", "-i", str(abl / pre)], cwd=work, check=True, capture_output=True)
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", str(abl / name)], cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, f"{name} does not apply to {e.get('base', man['base_commit'])}: {r.stdout[-400:]}"
