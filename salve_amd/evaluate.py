"""Inference drivers and the prediction wire format (scripts/test.py of the reference).

* `save_edge_classifications_to_disk` -- one `batch_{i}.json` per batch with y_hat, y_true, y_hat_probs, fp0, fp1
  (scripts/test.py:52-81); fp0 / fp1 are tile paths whose NAMES carry the hypothesis identity that
  salve/common/edge_classification.py:145-175 parses back (building = parent directory, floor id, pano ids,
  identity|rotated, W/D/O pair uuid), so the pose-graph stage consumes GPU results unchanged.
* `run_test_epoch` / `evaluate_model` -- the un-fused path: tiles from disk (dataset/zind_data.py) through the verifier
  (scripts/test.py:155-303), minus the matplotlib visualisation.
* `run_fused_epoch` -- the MI355X path: hypotheses -> rasteriser -> verifier in HBM (pipeline.py), same prediction files,
  no JPEG in between.
"""

from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from salve_amd.utils import pr_utils


class PrecisionRecallMeter:
    """Running precision / recall / mean accuracy over batches (scripts/test.py:25-49)."""

    def __init__(self) -> None:
        self.all_y_true = np.zeros((0, 1))
        self.all_y_hat = np.zeros((0, 1))

    def update(self, y_true: np.ndarray, y_hat: np.ndarray) -> None:
        self.all_y_true = np.vstack([self.all_y_true, np.asarray(y_true).reshape(-1, 1)])
        self.all_y_hat = np.vstack([self.all_y_hat, np.asarray(y_hat).reshape(-1, 1)])

    def get_metrics(self):
        return pr_utils.compute_precision_recall(y_true=self.all_y_true, y_pred=self.all_y_hat)


class ClassAccuracyMeter:
    """Per-class accuracy and its mean: what the reference reads off SegmentationAverageMeter (avg_meter.py:35-110:
    intersection / target counts per class, epsilon 1e-10)."""

    def __init__(self, num_classes: int) -> None:
        self.correct = np.zeros(num_classes)
        self.total = np.zeros(num_classes)

    def update(self, pred: np.ndarray, target: np.ndarray) -> None:
        pred, target = np.asarray(pred).reshape(-1), np.asarray(target).reshape(-1)
        for c in range(len(self.total)):
            self.total[c] += (target == c).sum()
            self.correct[c] += np.logical_and(target == c, pred == c).sum()

    def get_metrics(self):
        accs = self.correct / (self.total + 1e-10)
        return accs, float(np.mean(accs))


def save_edge_classifications_to_disk(serialization_save_dir: str, batch_idx: int, y_hat: torch.Tensor, y_true: torch.Tensor,
                                      probs: torch.Tensor, fp0: Sequence[str], fp1: Sequence[str]) -> None:
    """scripts/test.py:52-81: `y_hat_probs` is the probability of the PREDICTED class; JSON with indent 4
    (salve/utils/io.py:24-36)."""
    n = y_hat.shape[0]
    save_dict = {
        "y_hat": y_hat.cpu().numpy().tolist(),
        "y_true": y_true.cpu().numpy().tolist(),
        "y_hat_probs": probs[torch.arange(n, device=probs.device), y_hat].cpu().numpy().tolist(),
        "fp0": list(fp0),
        "fp1": list(fp1),
    }
    os.makedirs(serialization_save_dir, exist_ok=True)
    with open(f"{serialization_save_dir}/batch_{batch_idx}.json", "w") as f:
        json.dump(save_dict, f, indent=4)


def _summary(split: str, ckpt_fpath: str, cls: ClassAccuracyMeter, pr: PrecisionRecallMeter) -> Dict[str, Any]:
    accs, avg = cls.get_metrics()
    prec, rec, macc = pr.get_metrics()
    return {"split": split, "checkpoint_file_path": ckpt_fpath, "average_accuracy": avg, "class_accuracies": accs.tolist(),
            "precision": prec, "recall": rec, "mean_accuracy": macc}


def batch_block(n_examples: int, batch_size: int, rank: int, world: int):
    """(first example, one past the last example, first batch index, examples per rank [world]) of rank `rank` when the
    ceil(n / batch_size) batches of an un-shuffled epoch are split into `world` contiguous blocks of WHOLE batches: batch
    boundaries -- and with them the `batch_{i}.json` files -- are those of the single-process run."""
    from salve_amd.synthetic import HypothesisTable

    nb = -(-n_examples // batch_size) if n_examples else 0
    bounds = [HypothesisTable.shard_bounds(nb, r, world) for r in range(world)]
    counts = [max(0, min(hi * batch_size, n_examples) - lo * batch_size) for lo, hi in bounds]
    blo, bhi = bounds[rank]
    return blo * batch_size, min(bhi * batch_size, n_examples), blo, counts


def sharded_loader(dataset, batch_size: int, rank: int, world: int):
    """The un-fused driver's multi-GPU decomposition (the reference: nn.DataParallel scatters every batch and gathers the model
    outputs, train_utils.py:214-215): one process per GPU instead, rank r reads and scores a contiguous block of whole batches.
    Returns (DataLoader over the rank's examples, index of its first batch, examples per rank)."""
    lo, hi, first_batch, counts = batch_block(len(dataset), batch_size, rank, world)
    part = dataset if world == 1 else torch.utils.data.Subset(dataset, range(lo, hi))
    return torch.utils.data.DataLoader(part, batch_size=batch_size, shuffle=False, num_workers=0, drop_last=False), first_batch, counts


@torch.no_grad()
def run_test_epoch(args, serialization_save_dir: str, ckpt_fpath: str, model, data_loader, split: str, save_viz: bool = False,
                   serialize_predictions: bool = True, world: int = 1, rank: int = 0, first_batch: int = 0,
                   counts: Optional[Sequence[int]] = None) -> Dict[str, Any]:
    """scripts/test.py:155-277.  Batches are (x1, x2[, x3, x4[, x5, x6]], is_match, fp0, fp1).
    world > 1 (`sharded_loader`): `data_loader` holds this rank's block of whole batches, `first_batch` the index of its first
    one (every rank writes its own `batch_{i}.json` files -- the same files a single process writes), `counts` the examples per
    rank; the predictions are collected with the path's ONE all-gather (pipeline.gather_logits) and every rank returns the
    metrics of the whole split."""
    from salve_amd import train_utils

    if save_viz:
        raise RuntimeError("false-positive visualisation (matplotlib) is outside the accelerated path")
    cls, pr = ClassAccuracyMeter(args.num_ce_classes), PrecisionRecallMeter()
    mine = []   # (y_hat, y_true) of this rank's examples, for the gather
    dev = None
    for i, example in enumerate(data_loader, start=first_batch):
        *xs, is_match, fp0, fp1 = example
        xs = list(xs) + [None] * (6 - len(xs))
        dev = xs[0].device if not torch.cuda.is_available() else torch.device("cuda")
        xs = [x.to(dev, non_blocking=True) if x is not None else None for x in xs]
        gt = torch.as_tensor(is_match).to(dev)
        probs, _ = train_utils.cross_entropy_forward(model, split, *xs, gt)
        y_hat = torch.argmax(probs, dim=1)
        if dev.type == "cuda":   # the host reads the predictions next: saturated fp16 activations must not pass as logits
            from salve_amd import status

            status.check(dev, f"run_test_epoch, batch {i}")
        if world > 1:
            mine.append(torch.stack([y_hat.float(), gt.reshape(-1).float()], 1))
        else:
            cls.update(y_hat.cpu().numpy(), gt.reshape(-1).cpu().numpy())
            pr.update(y_true=gt.reshape(-1).cpu().numpy(), y_hat=y_hat.cpu().numpy())
        if serialize_predictions:
            save_edge_classifications_to_disk(serialization_save_dir, i, y_hat, gt.reshape(-1), probs, fp0, fp1)
    if world > 1:
        from salve_amd.pipeline import gather_logits

        if counts is None:
            raise RuntimeError("run_test_epoch with world > 1 needs `counts` (evaluate.sharded_loader)")
        if dev is None:   # a rank without a batch still takes part in the collective
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        local = torch.cat(mine, 0) if mine else torch.zeros((0, 2), dtype=torch.float32, device=dev)
        both = gather_logits(local, world, counts=list(counts)).cpu().numpy()
        cls.update(both[:, 0].astype(np.int64), both[:, 1].astype(np.int64))
        pr.update(y_true=both[:, 1].astype(np.int64), y_hat=both[:, 0].astype(np.int64))
    return _summary(split, ckpt_fpath, cls, pr)


def evaluate_model(serialization_save_dir: str, ckpt_fpath: str, args, split: str, save_viz: bool = False, world: int = 1,
                   rank: int = 0) -> Dict[str, Any]:
    """scripts/test.py:280-303: data loader -> model -> checkpoint -> run_test_epoch -> `{ckpt stem}.json` summary.
    world > 1: one process per GPU (torch.distributed initialised by the caller), every rank scores its block of whole batches
    (`sharded_loader`), rank 0 writes the summary."""
    from salve_amd import train_utils

    model = train_utils.load_model_checkpoint(ckpt_fpath, train_utils.get_model(args), args)
    if world == 1:
        loader = train_utils.get_dataloader(args, split=split)
        metrics = run_test_epoch(args, serialization_save_dir, ckpt_fpath, model.eval(), loader, split, save_viz)
    else:
        from salve_amd.dataset.zind_data import ZindData

        data = ZindData(split=split, transform=train_utils.get_img_transform_list(args, split), args=args)
        loader, first_batch, counts = sharded_loader(data, args.batch_size, rank, world)
        metrics = run_test_epoch(args, serialization_save_dir, ckpt_fpath, model.eval(), loader, split, save_viz, world=world, rank=rank,
                                 first_batch=first_batch, counts=counts)
    if rank == 0:
        with open(f"{Path(ckpt_fpath).stem}.json", "w") as f:
            json.dump(metrics, f, indent=4)
    return metrics


@torch.no_grad()
def run_fused_epoch(pipe, hypotheses, tile_names: Sequence[Sequence[str]], y_true: Optional[np.ndarray], serialization_save_dir: str,
                    batch_size: int = 64, ckpt_fpath: str = "", split: str = "test", world: int = 1, rank: int = 0) -> Dict[str, Any]:
    """Render + verify a hypothesis table on the GPU (pipeline.RenderVerifyPipeline) and write the SAME prediction
    files as run_test_epoch.  `hypotheses`, `tile_names` and `y_true` describe the WHOLE table on every rank;
    `tile_names[j]` = (fp0, fp1): the tile paths the un-fused path would have written for hypothesis j
    (utils/bev_rendering_utils.bev_fname_from_img_fpath under `{root}/{label}/{building}/`, in file-name order); they are
    only used as names.  y_true: [N] labels (0 / 1) or None (then 0).  With world > 1 every rank scores its contiguous
    block (HypothesisTable.shard) and rank 0 writes the files after the path's one all-gather.

    Hypotheses with a render that has no point inside the BEV window are dropped: the reference writes no tile for them
    (bev_rendering_utils.py:279-280, 623-627), so they never reach scripts/test.py nor the pose-graph stage."""
    from salve_amd.pipeline import gather_logits

    n_all = len(hypotheses)
    shard = hypotheses.shard(rank, world) if world > 1 else hypotheses
    prepared = pipe.prepare(shard)
    local = pipe.score(prepared)
    valid_local = torch.from_numpy(pipe.valid_mask(prepared).astype(np.float32)).to(local.device)
    pipe.check("run_fused_epoch")
    # one collective: the validity flag rides along as an extra column of the logits
    both = gather_logits(torch.cat([local, valid_local[:, None]], dim=1), world, total=n_all)
    logits, valid = both[:, :-1], both[:, -1] > 0.5
    keep = torch.nonzero(valid).reshape(-1)
    probs = torch.softmax(logits[keep], dim=1)  # train_utils.py:31
    y_hat = torch.argmax(probs, dim=1)
    n = int(probs.shape[0])
    gt_all = torch.zeros(n_all, dtype=torch.long, device=probs.device) if y_true is None else torch.as_tensor(np.asarray(y_true), device=probs.device).long()
    gt = gt_all[keep]
    cls, pr = ClassAccuracyMeter(int(logits.shape[1])), PrecisionRecallMeter()
    cls.update(y_hat.cpu().numpy(), gt.cpu().numpy())
    pr.update(y_true=gt.cpu().numpy(), y_hat=y_hat.cpu().numpy())
    if rank == 0:
        kept = keep.cpu().numpy().tolist()
        for b, lo in enumerate(range(0, n, batch_size)):
            hi = min(lo + batch_size, n)
            save_edge_classifications_to_disk(serialization_save_dir, b, y_hat[lo:hi], gt[lo:hi], probs[lo:hi],
                                              [tile_names[kept[j]][0] for j in range(lo, hi)], [tile_names[kept[j]][1] for j in range(lo, hi)])
    out = _summary(split, ckpt_fpath, cls, pr)
    out["num_hypotheses"] = n_all
    out["num_dropped_no_points_in_window"] = n_all - n
    return out
