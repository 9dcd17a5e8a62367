"""RCCL on the one GPU a test box has: bench.py under torch.distributed.run with ONE rank and --force-dist.

The multi-GPU decomposition of the path (contiguous hypothesis shards, one all-gather of the logits; SURVEY 8e, reference
train_utils.py:214-215) is covered with gloo on the CPU (tests/test_distributed_cpu.py).  What gloo cannot show is that the
RCCL calls themselves work on this stack: `init_process_group("nccl", device_id=...)`, the `all_gather_into_tensor` of the
logits on DEVICE memory, the barrier / max-reduce of the timing and `destroy_process_group`.  This test runs exactly the
command line the driver uses for N > 1, with N = 1: the only thing `--gpus 8` adds is the number of ranks.

The launcher is started as a CHILD process (the parent keeps running; nothing is exec'ed in place)."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_the_launcher_with_rccl_on_one_gpu():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--hyps", "256",
           "--panos", "8", "--force-dist", "--no-cpu-baseline", "--no-calibration", "--no-config5"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["rccl"] is True
    assert out["config"]["hypotheses_per_gpu"] == 256 and out["value"] > 0
    assert out["roofline"]["renders_per_launch"] == 256 and out["roofline_verifier"]["samples_per_launch"] == 256
    print(f"RCCL world of one: {out['value']:.0f} hypotheses/s at 256 hypotheses per launch")


_PADDED_GATHER = '''
import sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from salve_amd.pipeline import gather_logits
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
g = torch.Generator().manual_seed(1)
local = torch.randn((4097, 2), generator=g).to(dev)
# a block that is SHORTER than ceil(total / world): the rank pads it to 4100 rows for the single all_gather_into_tensor
out = gather_logits(local, 1, total=4100, force=True)
torch.cuda.synchronize()
assert tuple(out.shape) == (4100, 2), out.shape
assert torch.equal(out[:4097], local) and not out[4097:].any()
same = gather_logits(local, 1, total=4097, force=True)      # equal blocks: no padding, same collective
assert torch.equal(same, local)
dist.destroy_process_group()
print("PADDED_GATHER_OK")
'''


def test_padded_block_gather_runs_on_rccl(tmp_path):
    """pipeline.gather_logits pads a rank's block to ceil(total / world) rows when the table does not divide evenly (gloo tests:
    tests/test_distributed_cpu.py).  The same code path -- zero-padded buffer, all_gather_into_tensor on DEVICE memory, the
    padding sliced off by shard bounds -- on RCCL, in a world of one whose block is shorter than the stated total (4097 of 4100
    rows: the odd table of an 8-GPU run seen from one rank).  Reference: the DataParallel gather of train_utils.py:214-215."""
    script = tmp_path / "padded_gather.py"
    script.write_text(_PADDED_GATHER.format(root=str(ROOT)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert proc.returncode == 0 and "PADDED_GATHER_OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]
