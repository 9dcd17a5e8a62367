"""Model build / checkpoint load / forward / data-loading helpers with the reference's names.

Mirror of the inference-side functions of salve/train_utils.py: get_model (:205-217), load_model_checkpoint
(:229-242), cross_entropy_forward (:18-41), get_val_test_transform (:126-159), get_img_transform_list (:162-170),
get_dataloader (:183-203).  Training-only helpers (optimiser, augmentation, LR schedule) are out of scope of the
accelerated path.
"""

from __future__ import annotations

from pathlib import Path
from typing import Tuple

import torch
from torch import Tensor, nn

from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.training_config import TrainingConfig


def get_model(args: TrainingConfig) -> nn.Module:
    """EarlyFusionCEResnet on the GPU.  `args.dataparallel` is accepted for config compatibility: the MI355X path is
    one process per GPU with a full replica each (salve_amd/pipeline.py), not nn.DataParallel."""
    model = EarlyFusionCEResnet(args.num_layers, args.pretrained, args.num_ce_classes, args)
    if torch.cuda.is_available():
        model = model.cuda()
    return model.eval()


def load_model_checkpoint(ckpt_fpath: str, model: nn.Module, args: TrainingConfig) -> nn.Module:
    """`checkpoint["state_dict"]`, strict.  Checkpoints saved from nn.DataParallel carry a `module.` prefix
    (scripts/train.py:97-107 with dataparallel: True in every released config); it is stripped."""
    if not Path(ckpt_fpath).exists():
        raise RuntimeError(f"=> no checkpoint found at {ckpt_fpath}")
    checkpoint = torch.load(ckpt_fpath, map_location="cpu", weights_only=False)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint["state_dict"].items()}
    model.load_state_dict(sd, strict=True)
    return model


def cross_entropy_forward(model: nn.Module, split: str, x1: Tensor, x2: Tensor, x3: Tensor, x4: Tensor, x5: Tensor,
                          x6: Tensor, is_match: Tensor) -> Tuple[Tensor, Tensor]:
    """(softmax probabilities, cross-entropy loss).  Inference only: split == "train" is refused."""
    if split == "train":
        raise RuntimeError("the HIP verifier is inference-only; training is outside the accelerated path")
    with torch.no_grad():
        logits = model(x1, x2, x3, x4, x5, x6)
        probs = torch.nn.functional.softmax(logits.clone(), dim=1)
        loss = torch.nn.functional.cross_entropy(logits, is_match.squeeze())
    return probs, loss


def get_val_test_transform(args: TrainingConfig):
    """Resize(resize_h, resize_w) -> centre Crop(train_h, train_w) -> ToTensor -> Normalize(ImageNet), for 1, 2 or 3
    modalities (2 / 4 / 6 images per example): salve/train_utils.py:126-159.  Runs on the GPU (transforms.py)."""
    from salve_amd.transforms import ValTestTransform

    if len(args.modalities) not in (1, 2, 3):
        raise RuntimeError(f"Unsupported modalities. {str(args.modalities)}")
    return ValTestTransform((args.resize_h, args.resize_w), (args.train_h, args.train_w))


def get_img_transform_list(args: TrainingConfig, split: str):
    if split == "train":
        raise RuntimeError("training augmentation is outside the accelerated path")
    if split not in ("val", "test"):
        raise RuntimeError(f"unknown split {split}")
    return get_val_test_transform(args)


def get_dataloader(args: TrainingConfig, split: str) -> torch.utils.data.DataLoader:
    """DataLoader over the rendered tiles of a split (salve/train_utils.py:183-203): no shuffling, no dropped batch for
    val / test.  The transform launches GPU kernels, so tiles are decoded in the calling process (num_workers = 0)
    instead of in `args.workers` forked workers."""
    from salve_amd.dataset.zind_data import ZindData

    data = ZindData(split=split, transform=get_img_transform_list(args, split), args=args)
    return torch.utils.data.DataLoader(data, batch_size=args.batch_size, shuffle=False, num_workers=0, drop_last=False)
