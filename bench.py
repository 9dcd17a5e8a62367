"""Benchmark of the SALVe hot path: alignment hypotheses / second (render + verify) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the fused render+verify path over the rank's shard of the hypothesis table
(BASELINE.json configs[2]: 4096 hypotheses over 64 synthetic 1024x512 panoramas, rasteriser + ResNet-50 fp16,
per GPU -> weak scaling).  Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

# algorithmic bytes of one render at 1024x512 -> 501x501
PANO_H, PANO_W, CROP = 512, 1024, 80
BYTES_PER_RENDER = (PANO_H - 2 * CROP) * PANO_W * (3 + 2) + 501 * 501 * 3  # RGB u8 + depth u16 read, BEV u8 written
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes of one bev_densify_kernel launch of 512 renders from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate
# passes, KiB units; profiles/r01_pmc_traffic.md).  The kernel reads 4 B / lane, for which FETCH_SIZE is uncalibrated:
# the read side is taken as counted (lower bound).
DENSIFY_TRAFFIC_BYTES_PER_RENDER = (472007 + 937658) * 1024 / 512  # measured on launches of 512 renders; one workgroup per render


def cpu_baseline(n_hyp: int, procs: int):
    """The oracle's reference-faithful path (scipy griddata + torch-CPU ResNet-50 fp32) on the host cores,
    `procs` worker processes (the reference's own parallelism is a multiprocessing.Pool,
    scripts/render_dataset_bev.py:111-113), on a bounded sample of the same workload."""
    import multiprocessing as mp

    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_cpu_one, list(range(n_hyp)))
    dt = time.time() - t0
    return n_hyp / dt, dt


def _cpu_one(i: int) -> int:
    torch.set_num_threads(1)
    from oracle import bev_oracle as bo
    from oracle import resnet_oracle as ro
    from salve_amd import synthetic
    from salve_amd.models.early_fusion import EarlyFusionCEResnet

    hyp = synthetic.make_hypotheses(64, 64, seed=0)
    p1, p2 = synthetic.make_pano(int(hyp.i1[i])), synthetic.make_pano(int(hyp.i2[i]))
    r1, r2 = bo.render_bev_pair(p1[0], p1[1], p2[0], p2[1], hyp.R[i], hyp.t[i], "floor", mode="scipy")
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"]))
    if r1 is not None:
        x1 = torch.from_numpy(bo.tile_from_bev(r1["bev"]))[None]
        x2 = torch.from_numpy(bo.tile_from_bev(r2["bev"]))[None]
        with torch.no_grad():
            ro.forward(model.state_dict(), 50, [x1, x2])
    return i


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hyps", type=int, default=4096, help="hypotheses per GPU")
    ap.add_argument("--panos", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=1024, help="hypotheses per render / verify launch")
    ap.add_argument("--no-overlap", action="store_true", help="render and verify on one HIP stream")
    ap.add_argument("--streams", type=int, default=3, help="2: rasteriser | verifier; 3: scatter | densify | verifier")
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    from salve_amd import synthetic
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline, gather_logits

    torch.manual_seed(0)
    model = EarlyFusionCEResnet(args.layers, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    pipe = RenderVerifyPipeline(model, dev, pano_hw=(PANO_H, PANO_W), chunk=args.chunk, overlap=not args.no_overlap, streams=args.streams)
    panos = [synthetic.make_pano(i, PANO_H, PANO_W) for i in range(args.panos)]
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    # weak scaling: every rank scores its own block of `hyps` hypotheses out of a table of world * hyps
    table = synthetic.make_hypotheses(args.hyps * world, args.panos, seed=0).shard(rank, world)
    prepared = pipe.prepare(table)
    logits = torch.empty((len(table), 2), dtype=torch.float32, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.score(prepared, out=logits)
        gather_logits(logits, world)
    stream = torch.cuda.current_stream(dev)
    ev = []  # (start, end, renders) of every bev_densify_kernel launch of the timed region
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.score(prepared, out=logits, timers=ev)
        allg = gather_logits(logits, world)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        n_total = args.hyps * world
        value = n_total * args.steps / dt
        # dominant kernel: bev_densify_kernel, one launch = `chunk` renders; HIP events on the launch stream bracket every
        # launch of the timed region (with the two-stream default they include the slow-down from the verifier's kernels
        # sharing the CUs, exactly as the rocprofv3 kernel trace of the same command does)
        full = [(a.elapsed_time(b), r) for a, b, r in ev if r == min(args.chunk, len(table))]
        dens_ms = float(np.mean([t for t, _ in full]))
        renders = full[0][1]
        achieved = renders * BYTES_PER_RENDER / (dens_ms * 1e-3) / 1e9
        out = {
            "metric": "alignment hypotheses/sec (render+verify)", "value": round(value, 2), "unit": "hypotheses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "config": {"workload": f"{args.hyps} hypotheses/GPU over {args.panos} synthetic 1024x512 panoramas, floor surface, "
                                   f"HIP BEV rasteriser + ResNet-{args.layers} (6-ch early fusion) fp16 MFMA verifier",
                       "hypotheses_per_gpu": args.hyps, "panos": args.panos, "renders_per_hypothesis": 1,
                       "cached_identity_renders": args.panos, "chunk": args.chunk, "parallelism": f"hypothesis-shard x{world}"},
            "roofline": {"kernel": "bev_densify_kernel", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": int(DENSIFY_TRAFFIC_BYTES_PER_RENDER * renders),
                         "launch_ms": round(dens_ms, 3), "launches_timed": len(full), "renders_per_launch": renders, "algorithmic_bytes_per_render": BYTES_PER_RENDER},
        }
        if world == 1 and not args.no_cpu_baseline:
            procs = min(os.cpu_count() or 1, 8)
            v, secs = cpu_baseline(6 * procs, procs)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "hypotheses/s", "cores": procs, "kind": "port",
                                   "sample": f"{6 * procs} hypotheses of the same table (2 renders + ResNet-50 fp32 each), oracle scipy mode, "
                                             f"{procs} processes, {secs:.1f} s"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
