"""Hyper-parameter record of one experiment.  Mirror of salve/training_config.py:7-64 (same 24 fields, mutable),
plus a PyYAML loader standing in for hydra's `instantiate(cfg.TrainingConfig)` (scripts/test.py:372-375)."""

from dataclasses import dataclass
from typing import Optional, Tuple


@dataclass(frozen=False)
class TrainingConfig:
    lr_annealing_strategy: str
    base_lr: float
    weight_decay: float
    num_ce_classes: int
    print_every: int
    poly_lr_power: float
    optimizer_algo: str
    num_layers: int
    pretrained: bool
    dataparallel: bool
    resize_h: int
    resize_w: int
    train_h: int
    train_w: int
    apply_photometric_augmentation: bool
    modalities: Tuple[str]
    cfg_stem: str
    num_epochs: int
    workers: int
    batch_size: int
    data_root: str
    layout_data_root: str
    model_save_dirpath: str
    gpu_ids: Optional[str] = None


def load_training_config(yaml_fpath: str) -> TrainingConfig:
    """Read one of the reference's salve/configs/*.yaml files ({TrainingConfig: {_target_: ..., fields...}})."""
    import yaml

    with open(yaml_fpath, "r") as f:
        d = dict(yaml.safe_load(f)["TrainingConfig"])
    d.pop("_target_", None)
    d["modalities"] = tuple(d["modalities"])
    return TrainingConfig(**d)
