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
    full = gather_logits(local, world)
    torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_shard_and_all_gather(tmp_path):
    n, world = 64, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    table = synthetic.make_hypotheses(n, 5, seed=0)
    expect = torch.from_numpy(np.stack([table.theta_deg, table.t[:, 0].astype(np.float64)], 1)).float()
    for r in range(world):
        assert torch.equal(torch.load(tmp_path / f"r{r}.pt"), expect)


def test_shards_partition_the_table():
    table = synthetic.make_hypotheses(4096, 64, seed=0)
    for world in (1, 2, 4, 8):
        sizes = [len(table.shard(r, world)) for r in range(world)]
        assert sizes == [4096 // world] * world
