"""python -m salve_amd.check_checkpoint <train_ckpt.pth> [--layers 152] [--modalities ceiling_rgb_texture,floor_rgb_texture] [--tiles DIR] [-n 64]

What fp16 storage costs THIS checkpoint: loads a reference-format checkpoint (`{"state_dict": ...}`, optional `module.` prefix --
scripts/train.py:97-107, loaded strictly as salve/train_utils.py:229-242 does), runs N tile sets through the HIP engine (fp16
weights and activations, fp32 accumulation) and through a torch float32 evaluation of the SAME module tree on the SAME fp16-rounded
tiles (this package's own resnet_factory modules, eval-mode BatchNorm; not `oracle/`), and prints where the logits live and how far the
two are apart: |logit| range, absolute and relative logit error, error of the softmax probabilities, arg-max agreement.

The contract it checks (north_star asks for 1e-3 absolute):
    |logit error| <= 1e-3 x max(1, max |logit|)   and   |probability error| <= 1e-3   and   equal arg-max.
north_star's absolute 1e-3 holds for |logit| <= ~4 (ResNet-50) / ~2.3 (ResNet-152); beyond, the error is relative to the logit
(<= 7.7e-4 x |logit| measured, profiles/r05_logit_sweep.txt) while the probabilities -- what scripts/test.py:217-229 hands to its
consumers -- stay within 1e-3 absolute at any magnitude.  Exit code 0: contract met on every sample; 1: not met.

Tiles: `--tiles DIR` reads rendered BEV tiles (JPEG / PNG, the files scripts/render_dataset_bev.py writes), grouped in sorted
order into sets of 2 / 4 / 6 images and put through the val / test transform (train_utils.get_val_test_transform).  Without it the
tile sets are rendered here: seeded synthetic panoramas and random hypotheses through the fused pipeline -- mostly black images
with textured regions, the input the verifier actually sees.

This module is a CHECK beside the product path, never part of it: the float32 evaluation below exists only to be compared with.
"""

from __future__ import annotations

import argparse
import sys
from types import SimpleNamespace
from typing import List, Sequence

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


def float32_forward(model, xs: Sequence[torch.Tensor]) -> torch.Tensor:
    """The verifier's forward in plain torch float32 from the model's own modules (early_fusion.py:55-83: channel concatenation, the
    model's conv1, the trunk's bn1 / relu / maxpool / layer1-4 / avgpool, the model's fc; torchvision's block definitions:
    resnet_factory.py).  Comparison reference of this check only."""
    r = model.resnet

    def bn(x, m: nn.BatchNorm2d):
        return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)

    x = torch.cat(list(xs), dim=1)
    x = F.max_pool2d(F.relu(bn(F.conv2d(x, model.conv1.weight, None, 2, 3), r.bn1)), 3, 2, 1)
    for li in range(1, 5):
        for blk in getattr(r, f"layer{li}"):
            idn = x
            if blk.downsample is not None:
                idn = bn(F.conv2d(x, blk.downsample[0].weight, None, blk.downsample[0].stride), blk.downsample[1])
            if hasattr(blk, "conv3"):
                y = F.relu(bn(F.conv2d(x, blk.conv1.weight), blk.bn1))
                y = F.relu(bn(F.conv2d(y, blk.conv2.weight, None, blk.conv2.stride, 1), blk.bn2))
                y = bn(F.conv2d(y, blk.conv3.weight), blk.bn3)
            else:
                y = F.relu(bn(F.conv2d(x, blk.conv1.weight, None, blk.conv1.stride, 1), blk.bn1))
                y = bn(F.conv2d(y, blk.conv2.weight, None, 1, 1), blk.bn2)
            x = F.relu(y + idn)
    return F.linear(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1), model.fc.weight, model.fc.bias)


def compare(got: torch.Tensor, ref: torch.Tensor) -> dict:
    """Statistics of HIP logits `got` against float32 logits `ref` ([N, classes], CPU)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs().max(1).values
    mag = ref.abs().max(1).values
    perr = (torch.softmax(got, 1) - torch.softmax(ref, 1)).abs().max(1).values
    rel = err / mag.clamp(min=1.0)
    q = lambda t, p: float(t.quantile(p)) if t.numel() > 1 else float(t.max())
    return {"n": int(got.shape[0]), "logit_abs_min": float(mag.min()), "logit_abs_median": float(mag.median()), "logit_abs_max": float(mag.max()),
            "err_median": float(err.median()), "err_p99": q(err, 0.99), "err_max": float(err.max()),
            "rel_err_p99": q(rel, 0.99), "rel_err_max": float(rel.max()), "prob_err_max": float(perr.max()),
            "argmax_equal": int((got.argmax(1) == ref.argmax(1)).sum()),
            "abs_1e3_holds": bool(float(err.max()) <= 1e-3),
            "contract_holds": bool(float(rel.max()) <= 1e-3 and float(perr.max()) <= 1e-3 and bool((got.argmax(1) == ref.argmax(1)).all()))}


def report(st: dict, what: str) -> str:
    return (f"{what}: {st['n']} tile sets\n"
            f"  |logit| (largest per sample): min {st['logit_abs_min']:.2f}  median {st['logit_abs_median']:.2f}  max {st['logit_abs_max']:.2f}\n"
            f"  logit error, absolute:        median {st['err_median']:.2e}  p99 {st['err_p99']:.2e}  max {st['err_max']:.2e}"
            f"   (north_star's absolute 1e-3: {'holds' if st['abs_1e3_holds'] else 'does NOT hold at this magnitude'})\n"
            f"  error / max(1, |logit|):      p99 {st['rel_err_p99']:.2e}  max {st['rel_err_max']:.2e}   (contract: <= 1e-3)\n"
            f"  softmax probabilities:        max error {st['prob_err_max']:.2e}   (contract: <= 1e-3)\n"
            f"  arg-max equal:                {st['argmax_equal']} / {st['n']}\n"
            f"  contract {'MET' if st['contract_holds'] else 'NOT MET'}")


def _tiles_from_dir(tile_dir: str, n_images: int, n_sets: int, device) -> torch.Tensor:
    """fp16 NHWC tile sets [N, 224, 224, Cpad] from image files: sorted, grouped n_images at a time, through the val / test transform."""
    import glob
    import os

    from salve_amd.utils.image_io import read_rgb
    from salve_amd.models.hip_resnet import nchw_to_input, pad_channels
    from salve_amd.transforms import ValTestTransform

    files = sorted(f for ext in ("jpg", "jpeg", "png") for f in glob.glob(os.path.join(tile_dir, "**", f"*.{ext}"), recursive=True))
    if len(files) < n_images:
        raise SystemExit(f"check_checkpoint: {tile_dir} holds {len(files)} image(s); one tile set needs {n_images}")
    tf = ValTestTransform((234, 234), (224, 224))
    sets = []
    for lo in range(0, min(len(files) // n_images, n_sets) * n_images, n_images):
        imgs = [read_rgb(f) for f in files[lo:lo + n_images]]
        xs = tf(*imgs)
        sets.append(nchw_to_input([x[None].to(device) for x in xs[:n_images]], pad_channels(3 * n_images)))
    return torch.cat(sets)


def _rendered_tiles(model, n_sets: int, device) -> torch.Tensor:
    """fp16 NHWC tile sets rendered by the fused pipeline from seeded synthetic panoramas (cluttered scene) and random hypotheses."""
    from salve_amd import synthetic
    from salve_amd.pipeline import RenderVerifyPipeline

    pipe = RenderVerifyPipeline(model, device, chunk=None, overlap=False, streams=1, n_hypotheses=n_sets)
    P = 8
    panos = [synthetic.make_pano(i, scene="cluttered") for i in range(P)]
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    pipe.score(pipe.prepare(synthetic.make_hypotheses(n_sets, P, seed=5)))
    pipe.check("check_checkpoint: rendering the tile sets")
    return pipe.tile_bufs[0][:n_sets].clone()


def run(model, tiles: torch.Tensor, device, batch: int = 16, threads: int = 0) -> dict:
    """HIP logits against the float32 evaluation on the same tiles (fp16 NHWC [N, 224, 224, Cpad] on `device`)."""
    n_img = model.num_images
    with torch.no_grad():
        got = model.forward_nhwc(tiles.contiguous()).cpu()
        model.check(device, "check_checkpoint: HIP forward")
        cpu_model = model   # parameters live wherever the caller put them; evaluate in float32 on the CPU copy of the tiles
        x = tiles[..., :3 * n_img].float().cpu().permute(0, 3, 1, 2).contiguous()
        if threads:
            torch.set_num_threads(threads)
        sd_dev = next(model.parameters()).device
        if sd_dev.type != "cpu":
            import copy

            cpu_model = copy.deepcopy(model).cpu()
            cpu_model._compiled = None
        ref = torch.cat([float32_forward(cpu_model, [x[lo:lo + batch, 3 * k:3 * k + 3] for k in range(n_img)]) for lo in range(0, x.shape[0], batch)])
    return compare(got, ref)


def main(argv: List[str] = None) -> int:
    ap = argparse.ArgumentParser(prog="python -m salve_amd.check_checkpoint", description=__doc__.split("\n\n")[1])
    ap.add_argument("checkpoint", help="train_ckpt.pth: a dict with 'state_dict' (scripts/train.py:97-107)")
    ap.add_argument("--layers", type=int, default=152, help="num_layers of the config the checkpoint was trained with (released models: 152)")
    ap.add_argument("--modalities", default="ceiling_rgb_texture,floor_rgb_texture", help="comma-separated, as in the config (TrainingConfig.modalities)")
    ap.add_argument("--tiles", default=None, help="directory of rendered BEV tiles; default: render synthetic tile sets here")
    ap.add_argument("-n", type=int, default=64, help="tile sets to run")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)

    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.train_utils import load_model_checkpoint

    mods = [m.strip() for m in args.modalities.split(",") if m.strip()]
    cfg = SimpleNamespace(modalities=mods)
    model = EarlyFusionCEResnet(args.layers, False, 2, cfg).eval()
    load_model_checkpoint(args.checkpoint, model, cfg)          # strict; RuntimeError if the file is missing, as the reference
    dev = torch.device(args.device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise SystemExit("check_checkpoint: needs the HIP device (the engine under test has no CPU path)")
    tiles = _tiles_from_dir(args.tiles, model.num_images, args.n, dev) if args.tiles else _rendered_tiles(model, args.n, dev)
    st = run(model, tiles, dev)
    print(report(st, f"{args.checkpoint} (ResNet-{args.layers}, {model.num_images} images per set, "
                     f"{'tiles from ' + args.tiles if args.tiles else 'tile sets rendered from synthetic panoramas'})"))
    return 0 if st["contract_holds"] else 1


if __name__ == "__main__":
    sys.exit(main())
