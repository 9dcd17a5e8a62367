"""GPU side of SURVEY section 8f rows 1-2: the validation transform kernel on the reference's own fixture tiles, the
un-fused test epoch (tiles from disk -> verifier -> batch_{i}.json) and the fused epoch writing the same files."""

import json
import shutil
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import bev_oracle as bo  # noqa: E402
from oracle import resnet_oracle as ro  # noqa: E402
from salve_amd import evaluate, synthetic, train_utils  # noqa: E402
from salve_amd.training_config import TrainingConfig  # noqa: E402
from salve_amd.utils import bev_rendering_utils as bru  # noqa: E402
from salve_amd.utils import image_io  # noqa: E402

RENDERINGS = Path(__file__).resolve().parent / "golden" / "renderings"


def config(data_root: str, modalities, num_layers=18, batch_size=4) -> TrainingConfig:
    return TrainingConfig(lr_annealing_strategy="poly", base_lr=1e-3, weight_decay=1e-4, num_ce_classes=2, print_every=10,
                          poly_lr_power=0.9, optimizer_algo="adam", num_layers=num_layers, pretrained=False, dataparallel=False,
                          resize_h=234, resize_w=234, train_h=224, train_w=224, apply_photometric_augmentation=False,
                          modalities=tuple(modalities), cfg_stem="t", num_epochs=1, workers=8, batch_size=batch_size,
                          data_root=data_root, layout_data_root="", model_save_dirpath="")


def test_val_transform_on_reference_fixture_tiles():
    """Resize 234 -> centre crop 224 -> ToTensor -> Normalize on the four 501x501 JPEG tiles the reference ships:
    bit-exact against the oracle's restatement of the cv2 / transform.py arithmetic."""
    files = sorted((RENDERINGS / "gt_alignment_approx" / "1208").glob("*.jpg"))
    imgs = [image_io.read_rgb(str(f)) for f in files]
    assert len(imgs) == 4 and all(im.shape == imgs[0].shape for im in imgs)
    tf = train_utils.get_val_test_transform(config("", ["ceiling_rgb_texture", "floor_rgb_texture"]))
    out = tf(*imgs)
    assert len(out) == 4
    for im, t in zip(imgs, out):
        exp = bo.tile_from_bev(im)
        assert t.shape == (3, 224, 224) and t.dtype == torch.float32 and t.is_cuda
        assert np.array_equal(t.cpu().numpy(), exp)
    with pytest.raises(RuntimeError):
        train_utils.get_img_transform_list(config("", ["floor_rgb_texture"]), "train")


def test_unfused_test_epoch_writes_prediction_files(tmp_path):
    root = tmp_path / "bev"
    shutil.copytree(RENDERINGS / "gt_alignment_approx", root / "gt_alignment_approx")
    neg = root / "incorrect_alignment" / "1208"
    neg.mkdir(parents=True)
    for f in (RENDERINGS / "gt_alignment_approx" / "1208").glob("*.jpg"):   # the same tiles, flipped, as a negative pair
        image_io.write_jpeg(str(neg / f.name.replace("pair_58", "pair_3")), image_io.read_rgb(str(f))[::-1].copy())
    args = config(str(root), ["ceiling_rgb_texture", "floor_rgb_texture"], num_layers=18)
    torch.manual_seed(0)
    model = train_utils.get_model(args)
    # building 1208 belongs to the official TRAIN split, for which get_dataloader refuses (no augmentation here): build
    # the same loader get_dataloader builds for val / test, over the train-split tiles
    from salve_amd.dataset.zind_data import ZindData

    with pytest.raises(RuntimeError):
        train_utils.get_dataloader(args, "train")
    assert len(train_utils.get_dataloader(args, "test").dataset) == 0
    data = ZindData(split="train", transform=train_utils.get_val_test_transform(args), args=args)
    loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, shuffle=False, num_workers=0, drop_last=False)
    assert len(loader.dataset) == 2
    metrics = evaluate.run_test_epoch(args, str(tmp_path / "preds"), "ckpt.pth", model, loader, "test")
    with open(tmp_path / "preds" / "batch_0.json") as f:
        d = json.load(f)
    assert d["y_true"] == [1, 0] and len(d["y_hat"]) == 2
    assert [Path(p).name.split("___")[0] for p in d["fp0"]] == ["pair_58", "pair_3"]
    assert all("_floor_rgb_" in Path(p).name for p in d["fp0"] + d["fp1"])
    # probabilities against the fp32 oracle on oracle tiles
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for j, (c1, c2, f1, f2, _) in enumerate(loader.dataset.data_list):
        xs = [torch.from_numpy(bo.tile_from_bev(image_io.read_rgb(p)))[None] for p in (c1, c2, f1, f2)]
        logits = ro.forward(sd, 18, xs)
        probs = torch.softmax(logits, 1)[0]
        assert int(torch.argmax(probs)) == d["y_hat"][j] or abs(float(probs[0] - probs[1])) < 5e-3
        assert abs(float(probs[d["y_hat"][j]]) - d["y_hat_probs"][j]) < 5e-3
    assert set(metrics) == {"split", "checkpoint_file_path", "average_accuracy", "class_accuracies", "precision", "recall", "mean_accuracy"}


def test_fused_epoch_writes_the_same_format(tmp_path):
    from types import SimpleNamespace

    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    pipe = RenderVerifyPipeline(model, dev, chunk=8)
    panos = [synthetic.make_pano(i) for i in range(3)]
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    hyp = synthetic.make_hypotheses(10, 3, seed=4)
    names = []
    for j in range(10):
        uuid = f"door_{j}_0_identity"
        names.append(tuple(f"/bev/gt_alignment_approx/0003/" + bru.bev_fname_from_img_fpath(j, uuid, "floor", f"/z/0003/panos/floor_01_partial_room_01_pano_{int(i)}.jpg")
                           for i in (hyp.i1[j], hyp.i2[j])))
    y_true = np.arange(10) % 2
    metrics = evaluate.run_fused_epoch(pipe, hyp, names, y_true, str(tmp_path / "preds"), batch_size=4)
    files = sorted((tmp_path / "preds").glob("batch_*.json"))
    assert [f.name for f in files] == ["batch_0.json", "batch_1.json", "batch_2.json"]
    probs = torch.softmax(pipe.score(pipe.prepare(hyp)), 1).cpu().numpy()
    got = [json.load(open(f)) for f in files]
    y_hat = sum((g["y_hat"] for g in got), [])
    p_hat = sum((g["y_hat_probs"] for g in got), [])
    assert sum((g["y_true"] for g in got), []) == y_true.tolist()
    assert y_hat == probs.argmax(1).tolist()
    assert np.allclose(p_hat, probs.max(1), atol=1e-6)
    assert sum((g["fp0"] for g in got), []) == [n[0] for n in names]
    assert 0.0 <= metrics["mean_accuracy"] <= 1.0
