"""python -m salve_amd.check_checkpoint: the float32 evaluation it compares the HIP engine with, its statistics and
its checkpoint handling on the CPU; the whole tool on a seeded checkpoint file on the GPU."""
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import resnet_oracle as ro
from salve_amd import check_checkpoint as cc
from salve_amd import synthetic
from salve_amd.models.early_fusion import EarlyFusionCEResnet

ROOT = Path(__file__).resolve().parents[1]


def _seeded_checkpoint(path, layers, modalities, prefix=""):
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=modalities)).eval()
    synthetic.trained_looking_batchnorm(model)
    synthetic.trained_looking_head(model, 30.0)
    # the reference's checkpoint format (scripts/train.py:97-107); DataParallel's `module.` prefix in every released config
    torch.save({"epoch": 3, "state_dict": {prefix + k: v for k, v in model.state_dict().items()}, "optimizer": {}}, path)
    return model


@pytest.mark.parametrize("layers,modalities", [(18, ["layout"]), (34, ["floor_rgb_texture"]), (50, ["ceiling_rgb_texture", "floor_rgb_texture"])])
def test_float32_forward_is_the_published_network(layers, modalities):
    """The comparison reference of the tool is built from the package's own module tree; it must be the same function as the oracle's
    restatement of torchvision's ResNet v1.5 (which tests/test_oracle_hf_resnet.py pins against an independent implementation)."""
    torch.manual_seed(1)
    model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=modalities)).eval()
    synthetic.trained_looking_batchnorm(model, seed=3)
    g = torch.Generator().manual_seed(2)
    n = model.num_images
    xs = [torch.randn(2, 3, 64, 64, generator=g) for _ in range(n)]
    with torch.no_grad():
        a = cc.float32_forward(model, xs)
        b = ro.forward(model.state_dict(), layers, xs)
    assert a.shape == (2, 2)
    assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_compare_and_report():
    ref = torch.tensor([[4.0, -4.0], [0.2, 0.1], [-10.0, 10.0]])
    got = ref + torch.tensor([[2e-3, 0.0], [5e-4, 0.0], [0.0, 8e-3]])
    st = cc.compare(got, ref)
    assert st["n"] == 3 and st["argmax_equal"] == 3
    assert abs(st["err_max"] - 8e-3) < 1e-6 and not st["abs_1e3_holds"]
    assert abs(st["rel_err_max"] - 8e-4) < 1e-6            # 8e-3 at |logit| 10; 2e-3 at 4 = 5e-4; 5e-4 at <1 stays absolute
    assert st["prob_err_max"] <= 1e-3 and st["contract_holds"]
    text = cc.report(st, "x")
    assert "does NOT hold" in text and "contract MET" in text
    bad = cc.compare(ref + torch.tensor([[0.0, 0.0], [0.0, 0.2], [0.0, 0.0]]), ref)   # arg-max flips on the close pair
    assert not bad["contract_holds"] and bad["argmax_equal"] == 2


def test_checkpoint_file_round_trip_with_module_prefix(tmp_path):
    from salve_amd.train_utils import load_model_checkpoint

    src = _seeded_checkpoint(tmp_path / "train_ckpt.pth", 18, ["floor_rgb_texture"], prefix="module.")
    dst = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    load_model_checkpoint(str(tmp_path / "train_ckpt.pth"), dst, None)
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(RuntimeError):
        load_model_checkpoint(str(tmp_path / "missing.pth"), dst, None)


def test_cli_refuses_to_run_without_the_device(tmp_path):
    """No HIP device here: the tool says so in one line and exits non-zero (it never falls back to comparing torch with torch)."""
    if torch.cuda.is_available():
        pytest.skip("this is the CPU container's behaviour")
    _seeded_checkpoint(tmp_path / "c.pth", 18, ["floor_rgb_texture"])
    proc = subprocess.run([sys.executable, "-m", "salve_amd.check_checkpoint", str(tmp_path / "c.pth"), "--layers", "18", "--modalities", "floor_rgb_texture"],
                          capture_output=True, text=True, cwd=str(ROOT), timeout=300)
    assert proc.returncode != 0 and "needs the HIP device" in proc.stderr


@pytest.mark.gpu
def test_cli_on_a_seeded_checkpoint(tmp_path):
    """The whole tool: checkpoint file -> strict load -> tile sets rendered by the pipeline -> HIP engine against the float32 evaluation.
    ResNet-50, logits of several units (head x 30): the relative contract holds, the absolute 1e-3 is reported as what it is."""
    _seeded_checkpoint(tmp_path / "train_ckpt.pth", 50, ["floor_rgb_texture"], prefix="module.")
    proc = subprocess.run([sys.executable, "-m", "salve_amd.check_checkpoint", str(tmp_path / "train_ckpt.pth"), "--layers", "50",
                           "--modalities", "floor_rgb_texture", "-n", "24"], capture_output=True, text=True, cwd=str(ROOT), timeout=900)
    print(proc.stdout)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert "contract MET" in proc.stdout and "arg-max equal:                24 / 24" in proc.stdout


@pytest.mark.gpu
def test_cli_on_tile_files(tmp_path):
    """--tiles: JPEG tiles on disk (the reference's four fixture tiles, tests/golden/renderings) through the val / test transform."""
    _seeded_checkpoint(tmp_path / "c.pth", 18, ["floor_rgb_texture"])
    tiles = ROOT / "tests" / "golden" / "renderings"
    proc = subprocess.run([sys.executable, "-m", "salve_amd.check_checkpoint", str(tmp_path / "c.pth"), "--layers", "18", "--modalities", "floor_rgb_texture",
                           "--tiles", str(tiles), "-n", "2"], capture_output=True, text=True, cwd=str(ROOT), timeout=900)
    print(proc.stdout)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    assert "tiles from" in proc.stdout and "contract MET" in proc.stdout
