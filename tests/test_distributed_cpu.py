"""The multi-GPU decomposition on CPU: contiguous hypothesis shards + one all-gather of logits (gloo, world size 2)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from salve_amd import synthetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from salve_amd.pipeline import gather_logits

    table = synthetic.make_hypotheses(n, 5, seed=0)
    shard = table.shard(rank, world)
    # stand-in for the verifier: a deterministic function of the hypothesis, so the gathered order can be checked
    local = torch.from_numpy(np.stack([shard.theta_deg, shard.t[:, 0].astype(np.float64)], 1)).float()
    full = gather_logits(local, world, total=n)
    torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(64, 2), (65, 2), (64, 3), (32768, 8), (32771, 8)])
def test_shard_and_all_gather(tmp_path, n, world):
    """Equal and unequal shards (65 rows over 2 ranks, 64 over 3): the single all_gather_into_tensor runs on blocks padded to
    ceil(n / world) rows and the padding is dropped, so every rank ends with the table in its original order.  (32768, 8) is BASELINE
    config 4's table on its eight ranks; (32771, 8) the same with a row count eight does not divide.)"""
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    table = synthetic.make_hypotheses(n, 5, seed=0)
    expect = torch.from_numpy(np.stack([table.theta_deg, table.t[:, 0].astype(np.float64)], 1)).float()
    for r in range(world):
        assert torch.equal(torch.load(tmp_path / f"r{r}.pt"), expect)


class _StubPipeline:
    """Stands in for pipeline.RenderVerifyPipeline in the N > 1 driver test below (the real one needs the GPU): logits are
    a deterministic function of the hypothesis, hypotheses with t_x > 1.5 count as "no point inside the window"."""

    def prepare(self, shard):
        return shard

    def score(self, shard):
        return torch.from_numpy(np.stack([shard.t[:, 0], shard.t[:, 1]], 1).astype(np.float32))

    def valid_mask(self, shard):
        return shard.t[:, 0] <= 1.5

    def check(self, what):
        pass


def _epoch_worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from salve_amd import evaluate

    table = synthetic.make_hypotheses(n, 5, seed=1)
    names = [(f"/bev/gt_alignment_approx/0001/a_{j}.jpg", f"/bev/gt_alignment_approx/0001/b_{j}.jpg") for j in range(n)]
    m = evaluate.run_fused_epoch(_StubPipeline(), table, names, np.arange(n) % 2, os.path.join(out_dir, "preds"), batch_size=16, world=world, rank=rank)
    torch.save(m, os.path.join(out_dir, f"m{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(37, 2), (50, 3)])
def test_fused_epoch_driver_on_several_ranks(tmp_path, n, world):
    """evaluate.run_fused_epoch with world > 1 (gloo): every rank is handed the WHOLE table, scores its own block, the one
    all-gather carries logits and the validity flag, rank 0 writes the prediction files -- without the hypotheses whose
    renders had no point inside the window, in table order."""
    import json

    mp.spawn(_epoch_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    table = synthetic.make_hypotheses(n, 5, seed=1)
    keep = np.nonzero(table.t[:, 0] <= 1.5)[0]
    assert 0 < len(keep) < n
    files = sorted((tmp_path / "preds").glob("batch_*.json"), key=lambda f: int(f.stem.split("_")[1]))
    got = [json.load(open(f)) for f in files]
    assert len(files) == -(-len(keep) // 16)
    assert sum((g["fp0"] for g in got), []) == [f"/bev/gt_alignment_approx/0001/a_{j}.jpg" for j in keep]
    assert sum((g["y_true"] for g in got), []) == (keep % 2).tolist()
    assert sum((g["y_hat"] for g in got), []) == np.argmax(table.t[keep], 1).tolist()
    metrics = [torch.load(tmp_path / f"m{r}.pt", weights_only=False) for r in range(world)]
    assert all(m["num_hypotheses"] == n and m["num_dropped_no_points_in_window"] == n - len(keep) for m in metrics)
    assert all(m["mean_accuracy"] == metrics[0]["mean_accuracy"] for m in metrics)


def test_bench_starts_its_own_launcher_for_several_gpus(monkeypatch):
    """`python bench.py --gpus 8` without a launcher must start torch.distributed.run as a CHILD process (never exec, and
    before any GPU call) and relay its exit code."""
    import importlib
    import subprocess
    import sys

    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)   # an 8-GPU node (the guard below is tested on its own)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert cmd[-7].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not torch.cuda.is_initialized()


def test_bench_refuses_more_gpus_than_the_node_has(tmp_path):
    """`python bench.py --gpus 8` on a node with fewer than 8 devices: non-zero exit with ONE clear line, before a launcher is started and
    before any process initialises a GPU (this container shows none; a one-GPU box takes the same path for --gpus 8)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    proc = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert proc.returncode != 0
    lines = [l for l in (proc.stdout + proc.stderr).splitlines() if l.strip()]
    assert len(lines) == 1 and "--gpus 8" in lines[0] and "GPU(s)" in lines[0], lines
    assert "torch.distributed" not in proc.stderr and "Traceback" not in proc.stderr


def test_shards_partition_the_table():
    table = synthetic.make_hypotheses(4096, 64, seed=0)
    for world in (1, 2, 4, 8):
        sizes = [len(table.shard(r, world)) for r in range(world)]
        assert sizes == [4096 // world] * world
    odd = synthetic.make_hypotheses(4099, 64, seed=0)
    for world in (2, 3, 8):
        shards = [odd.shard(r, world) for r in range(world)]
        assert sum(len(s) for s in shards) == 4099 and max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        assert np.array_equal(np.concatenate([s.theta_deg for s in shards]), odd.theta_deg)


class _StubTiles(torch.utils.data.Dataset):
    """Stands in for dataset.zind_data.ZindData in the un-fused N > 1 driver test: (x1, x2, is_match, fp0, fp1) examples."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, j):
        g = torch.Generator().manual_seed(j)
        return torch.randn(3, 4, 4, generator=g), torch.randn(3, 4, 4, generator=g), j % 2, f"/bev/x/0001/a_{j}.jpg", f"/bev/x/0001/b_{j}.jpg"


class _StubVerifier(torch.nn.Module):
    def forward(self, x1, x2, x3, x4, x5, x6):
        return torch.stack([x1.mean((1, 2, 3)), x2.mean((1, 2, 3))], 1)


def _unfused_worker(rank, world, port, n, bs, out_dir):
    from types import SimpleNamespace

    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from salve_amd import evaluate

    loader, first_batch, counts = evaluate.sharded_loader(_StubTiles(n), bs, rank, world)
    m = evaluate.run_test_epoch(SimpleNamespace(num_ce_classes=2), os.path.join(out_dir, f"preds_w{world}"), "", _StubVerifier().eval(), loader, "test",
                                world=world, rank=rank, first_batch=first_batch, counts=counts)
    torch.save(m, os.path.join(out_dir, f"u{world}_{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,bs,world", [(38, 4, 2), (21, 8, 3), (6, 4, 3)])
def test_unfused_test_epoch_on_several_ranks(tmp_path, n, bs, world):
    """evaluate.run_test_epoch with world > 1 (gloo): the batches of the un-shuffled epoch are split into contiguous blocks of
    WHOLE batches, every rank writes its own `batch_{i}.json` files under the batch's global index, one all-gather collects
    (y_hat, y_true) for the metrics -- files and metrics identical to the single-process run (6 examples in batches of 4 over 3 ranks: one rank
    has no batch and still takes part in the collective)."""
    import json

    _unfused_worker(0, 1, 0, n, bs, str(tmp_path))
    mp.spawn(_unfused_worker, args=(world, _free_port(), n, bs, str(tmp_path)), nprocs=world, join=True)
    one = sorted((tmp_path / "preds_w1").glob("batch_*.json"))
    many = sorted((tmp_path / f"preds_w{world}").glob("batch_*.json"))
    assert [f.name for f in one] == [f.name for f in many] and len(one) == -(-n // bs)
    for a, b in zip(one, many):
        assert json.load(open(a)) == json.load(open(b)), a.name
    ref = torch.load(tmp_path / "u1_0.pt", weights_only=False)
    for r in range(world):
        assert torch.load(tmp_path / f"u{world}_{r}.pt", weights_only=False) == ref


def test_render_pairs_deals_the_floor_list_round_robin(tmp_path, monkeypatch):
    """render_dataset.render_pairs(rank, world): the reference's (building, floor) work list (scripts/render_dataset_bev.py:151-184:
    buildings sorted, 1348 skipped) dealt round robin to the ranks -- disjoint, complete, in list order; no collective."""
    from salve_amd import render_dataset as rd

    monkeypatch.setitem(rd.DATASET_SPLITS, "tiny", ["0007", "0003", "1348", "0005"])
    for bid, floors in (("0003", ["floor_01", "floor_02"]), ("0005", ["floor_01"]), ("0007", ["floor_00", "floor_01", "floor_02"]), ("1348", ["floor_01"])):
        for f in floors:
            (tmp_path / "hyp" / bid / f).mkdir(parents=True)
    whole = rd.floor_work_list(str(tmp_path / "hyp"), "tiny", None)
    assert whole == [("0003", "floor_01"), ("0003", "floor_02"), ("0005", "floor_01"), ("0007", "floor_00"), ("0007", "floor_01"), ("0007", "floor_02")]
    seen = []
    monkeypatch.setattr(rd, "render_building_floor_pairs", lambda *a, **k: seen.append((a[4], a[5])) or 2)
    per_rank = []
    for r in range(4):
        seen.clear()
        assert rd.render_pairs(1, "d", "b", "raw", str(tmp_path / "hyp"), None, ["rgb_texture"], "tiny", None, rank=r, world=4) == 2 * len(seen)
        per_rank.append(list(seen))
    assert per_rank == [whole[0::4], whole[1::4], whole[2::4], whole[3::4]]
    assert sorted(sum(per_rank, [])) == sorted(whole)
    with pytest.raises(ValueError):
        rd.render_pairs(1, "d", "b", "raw", str(tmp_path / "hyp"), None, ["rgb_texture"], "tiny", "0003")
    with pytest.raises(ValueError):
        rd.render_pairs(1, "d", "b", "raw", str(tmp_path / "hyp"), None, ["rgb_texture"], "tiny", None, rank=4, world=4)
