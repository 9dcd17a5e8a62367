"""Model build / checkpoint load / forward helpers with the reference's names.

Mirror of the inference-side functions of salve/train_utils.py: get_model (:205-217), load_model_checkpoint
(:229-242), cross_entropy_forward (:18-41).  Training-only helpers (optimiser, augmentation, LR schedule) are out of
scope of the accelerated path.
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
