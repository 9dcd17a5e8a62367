"""Benchmark of the SALVe hot path: alignment hypotheses / second (render + verify) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU under torch.distributed.run (RCCL).  When the script is started WITHOUT a launcher
(`python bench.py --gpus 8`), it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a
child process -- before anything touches the GPU -- and relays the child's JSON line and exit code.

A step = one pass of the fused render+verify path over the rank's shard of the hypothesis table
(BASELINE.json configs[2]: 4096 hypotheses over 64 synthetic 1024x512 panoramas, rasteriser + ResNet-50 fp16,
per GPU -> weak scaling).  Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

BASELINE.json configs[4] on one GPU (same JSON schema, `config.workload` names it):
    python bench.py --pano-hw 1024x2048 --surfaces floor,ceiling --layers 152 --hyps 1024 --panos 16 --chunk 512
`--force-dist` runs `init_process_group("nccl")` and the logits all-gather even with a single rank (RCCL on one GPU).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0    # same guide: measured float4 copy
MFMA_PEAK_TFLOPS = 2500.0      # same guide: dense fp16 / bf16 MFMA peak
# SURVEY 8d (2 x MAC, conv + fc), keyed by (layers, input channels of the early fusion); tests/test_oracle_structure.py pins them
GFLOP_PER_SAMPLE = {(50, 6): 8.410, (50, 12): 8.882, (152, 6): 23.259, (152, 12): 23.731, (18, 6): 3.6}
# Counted HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes at the benchmark's launch shape) is READ from the
# file the profile refresh writes -- never a constant in this script; a workload the file does not cover reports null.
TRAFFIC_FILE = ROOT / "profiles" / "traffic.json"


def bytes_per_render(pano_h: int, pano_w: int) -> int:
    """algorithmic bytes of one render -- RGB u8 + depth u16 of the un-cropped rows read, the 501 x 501
    BEV u8 written (2.555 MB at 1024x512, 7.96 MB at 2048x1024)."""
    crop = int(pano_h * (80 / 512))
    return (pano_h - 2 * crop) * pano_w * (3 + 2) + 501 * 501 * 3


def counted_traffic(key: str, field: str = "bytes_per_unit"):
    """(bytes per unit, source) of `key` from profiles/traffic.json, or (None, None).  bytes_per_unit = FETCH_SIZE x 2 + WRITE_SIZE
    (the guide's gfx950 correction of the read counter) for BOTH rooflines; bytes_per_unit_raw_fetch = with FETCH_SIZE as counted."""
    try:
        e = json.loads(TRAFFIC_FILE.read_text())[key]
        return float(e[field]), e["source"]
    except Exception:
        return None, None


def counted_issue(key: str):
    """The densify kernel's instruction-issue counters at the launch shape `key` names (profiles/traffic.json, written by the SQ pass
    of the profile refresh): the rasteriser's dominant kernel is bound by vector-instruction issue, not by HBM, so
    the line carries that bound beside the contractual HBM fraction.  None where the file has no entry."""
    try:
        return json.loads(TRAFFIC_FILE.read_text())[key]
    except Exception:
        return None


def _hwmon_of(device_index: int):
    """The sysfs hwmon directory (power1_input in microwatts, freq1_input in Hz) of the GPU torch calls cuda:<device_index>, matched by
    PCI address; None if sysfs does not show it."""
    import glob

    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        return None
    for dev in glob.glob("/sys/class/drm/card*/device"):
        try:
            if os.path.basename(os.path.realpath(dev)) != want:
                continue
            for h in glob.glob(dev + "/hwmon/hwmon*"):
                if os.path.exists(h + "/power1_input") and os.path.exists(h + "/freq1_input"):
                    return h
        except OSError:
            continue
    return None


def power_probe(fn, seconds: float = 3.0, device_index: int = 0):
    """Board power and shader clock while `fn` (one verifier forward) runs in a sustained loop, OUTSIDE the timed region: the forward runs
    at the board's power limit, so the clock the matrix pipes actually get -- not the 2.4 GHz the 2.5 PFLOP/s peak
    assumes -- is part of what the roofline fraction means.  Read from the GPU's hwmon files in sysfs by a thread while the loop runs (no
    child process); None where sysfs does not show the device."""
    import threading

    hw = _hwmon_of(device_index)
    if hw is None:
        return None
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(1.0)   # let the clocks settle under load
        while not stop.is_set():
            try:
                pw = int(open(hw + "/power1_input").read()) / 1e6
                ck = int(open(hw + "/freq1_input").read()) / 1e6
                samples.append((pw, ck))
            except Exception:
                return
            time.sleep(0.1)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    stop.set()
    th.join(timeout=5)
    if not samples:
        return None
    try:
        cap = int(open(hw + "/power1_cap").read()) / 1e6
    except Exception:
        cap = None
    return {"power_w": round(float(np.mean([a for a, _ in samples])), 1), "sclk_mhz": round(float(np.mean([b for _, b in samples])), 1),
            "power_cap_w": cap, "samples": len(samples)}


def _cores() -> int:
    """Physical cores this process may use (the box's share), for the CPU baseline."""
    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        phys = os.cpu_count() or 1
    try:
        phys = min(phys, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    # a one-GPU box of this pool is a 16-core share of its host: more workers than that only oversubscribe it
    return max(1, min(int(phys), int(os.environ.get("SALVE_CPU_BASELINE_CORES", "16"))))


def _cpu_render_pair(i: int):
    """One hypothesis of BASELINE config 1 on the CPU: both renders of the pair (the reference re-renders the identity
    pano for every hypothesis, bev_rendering_utils.py:453-455) with the oracle in its reference-faithful scipy mode."""
    torch.set_num_threads(1)
    from oracle import bev_oracle as bo
    from salve_amd import synthetic

    hyp = synthetic.make_hypotheses(16, 1, seed=0)
    rgb, depth = synthetic.make_pano(0)
    r1, r2 = bo.render_bev_pair(rgb, depth, rgb, depth, hyp.R[i], hyp.t[i], "floor", mode="scipy")
    if r1 is None or r2 is None:
        return None
    return np.concatenate([bo.tile_from_bev(r1["bev"]), bo.tile_from_bev(r2["bev"])], 0)


def _cpu_worker_warm(_):
    """Pool warm-up: the imports a worker needs (a spawned interpreter starts empty; its start-up is not the rasteriser's time)."""
    torch.set_num_threads(1)
    from oracle import bev_oracle as bo  # noqa: F401
    from salve_amd import synthetic  # noqa: F401

    return os.getpid()


def cpu_baseline():
    """BASELINE.json configs[0], literally: ONE 1024x512 synthetic panorama + depth, 16 hypotheses, CPU rasteriser (the
    oracle's scipy mode = the reference's own call sequence) in a multiprocessing.Pool of `cores` workers (the reference's
    parallelism, scripts/render_dataset_bev.py:111-113; its default is 15 processes) followed by torch-CPU ResNet-50 fp32 on
    the 16 tile pairs with `cores` threads (scripts/test.py runs batch 64).  Also the single-process per-render latency."""
    import multiprocessing as mp

    from oracle import bev_oracle as bo
    from oracle import resnet_oracle as ro
    from salve_amd import synthetic
    from salve_amd.models.early_fusion import EarlyFusionCEResnet

    cores = _cores()
    n_hyp = 16
    # spawn, not fork: this process has initialised the GPU by the time the baseline runs.  The workers are started and
    # have imported their modules BEFORE the clock starts (the reference's Pool forks from a warm interpreter).
    with mp.get_context("spawn").Pool(min(cores, n_hyp)) as pool:
        pool.map(_cpu_worker_warm, list(range(4 * min(cores, n_hyp))), chunksize=1)
        t0 = time.perf_counter()
        tiles = pool.map(_cpu_render_pair, list(range(n_hyp)))
        t_render = time.perf_counter() - t0
    # single-process per-render latency (SURVEY 8d i)
    hyp = synthetic.make_hypotheses(16, 1, seed=0)
    rgb, depth = synthetic.make_pano(0)
    a = bo.xyzrgb_from_arrays(depth, rgb, bo.floor_ceiling_z_range("floor"))
    a, _ = bo.pose_pair(a, a[:1], hyp.R[0], hyp.t[0])
    t1 = time.perf_counter()
    bo.render_bev_image(a, mode="scipy")
    t_single = time.perf_counter() - t1
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    kept = [t for t in tiles if t is not None]
    x = torch.from_numpy(np.stack(kept))
    torch.set_num_threads(cores)
    t2 = time.perf_counter()
    with torch.no_grad():
        ro.forward(model.state_dict(), 50, [x[:, :3], x[:, 3:]])
    t_verify = time.perf_counter() - t2
    total = t_render + t_verify
    return {"value": round(n_hyp / total, 4), "unit": "hypotheses/s", "cores": cores, "kind": "port",
            "sample": f"BASELINE config 1: 1 synthetic 1024x512 panorama, 16 hypotheses = 32 renders (oracle, scipy mode) in a pool of "
                      f"{min(cores, n_hyp)} processes {t_render:.1f} s + ResNet-50 fp32 torch-CPU batch {len(kept)} on {cores} threads {t_verify:.1f} s; "
                      f"single-process render latency {t_single:.2f} s",
            "render_s": round(t_render, 3), "verify_s": round(t_verify, 3), "single_render_s": round(t_single, 3)}


def _self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never will) and relay its output and exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hyps", type=int, default=4096, help="hypotheses per GPU")
    ap.add_argument("--panos", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=4096, help="hypotheses per render / verify launch (the whole shard: launches of 4096 renders /\n"
                    "samples run 10 % faster per unit than launches of 1024; the rasteriser of pass k + 1 runs under the verifier of pass k)")
    ap.add_argument("--no-overlap", action="store_true", help="same as --streams 1")
    ap.add_argument("--streams", type=int, default=1, help="1: one HIP stream (default: with whole-shard launches the overlapped schedules gain\n"
                    "1 % -- the kernels then share the CUs and each runs longer -- and blur the per-kernel times the rooflines are computed from);\n"
                    "2: rasteriser | verifier; 3: scatter | densify | verifier, the rasteriser of pass k + 1 under the verifier of pass k")
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--pano-hw", default="512x1024", help="panorama HxW: 512x1024 (configs 1-4) | 1024x2048 (config 5)")
    ap.add_argument("--surfaces", default="floor", help="floor | ceiling | floor,ceiling (config 5: 12-channel early fusion)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the logits all-gather even with one rank")
    ap.add_argument("--scene", default="box", help="synthetic scene: box (SURVEY 8d) | cluttered (occluding boxes + door opening) | noisy (cluttered + network-like depth errors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the 3 s loop of verifier forwards during which board power and shader clock are read from sysfs")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(_self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    pano_h, pano_w = (int(v) for v in args.pano_hw.lower().split("x"))
    surfaces = [v.strip() for v in args.surfaces.split(",")]
    modalities = [f"{v}_rgb_texture" for v in surfaces]
    S = len(surfaces)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:   # --force-dist without a launcher: a world of one on the loop-back interface
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    from salve_amd import synthetic
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline, gather_logits

    torch.manual_seed(0)
    model = EarlyFusionCEResnet(args.layers, False, 2, SimpleNamespace(modalities=modalities)).eval()
    synthetic.trained_looking_batchnorm(model)  # random-init weights of the named architecture; seeded trained-looking statistics
    pipe = RenderVerifyPipeline(model, dev, pano_hw=(pano_h, pano_w), chunk=args.chunk, overlap=(args.streams > 1 and not args.no_overlap), streams=args.streams)
    panos = [synthetic.make_pano(i, pano_h, pano_w, scene=args.scene) for i in range(args.panos)]
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    # weak scaling: every rank scores its own block of `hyps` hypotheses out of a table of world * hyps
    n_total = args.hyps * world
    table = synthetic.make_hypotheses(n_total, args.panos, seed=0).shard(rank, world)
    prepared = pipe.prepare(table)
    logits = torch.empty((len(table), 2), dtype=torch.float32, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.score(prepared, out=logits)
        gather_logits(logits, world, total=n_total, force=args.force_dist)
    ev, vev = [], []  # HIP events around every rasteriser stage / verifier forward of the timed region, on their own streams
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.score(prepared, out=logits, timers=ev, vtimers=vev)
        allg = gather_logits(logits, world, total=n_total, force=args.force_dist)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    pipe.check("bench.py")            # no star walk failed, no activation left the fp16 range
    assert allg.shape[0] == n_total and bool(torch.isfinite(allg).all())

    if rank == 0:
        value = n_total * args.steps / dt
        full_n = min(args.chunk, len(table))           # hypotheses per launch
        renders = full_n * S                           # renders per rasteriser launch
        bpr = bytes_per_render(pano_h, pano_w)
        mean_ms = lambda tag: float(np.mean([a.elapsed_time(b) for a, b, r, t in ev if t == tag and r == renders]))
        scat_ms, dens_ms = mean_ms("scatter"), mean_ms("densify")
        ras_ms = scat_ms + dens_ms
        achieved = renders * bpr / (ras_ms * 1e-3) / 1e9
        vfull = [(a.elapsed_time(b), r) for a, b, r in vev if r == full_n]
        ver_ms = float(np.mean([t for t, _ in vfull]))
        gflop = GFLOP_PER_SAMPLE[(args.layers, 6 * S)]
        tflops = full_n * gflop / ver_ms  # GFLOP / ms = TFLOP/s
        shape = f"{pano_w}x{pano_h}/{'+'.join(surfaces)}/resnet{args.layers}/launch{full_n}"
        ras_traffic, ras_src = counted_traffic(f"rasteriser/{pano_w}x{pano_h}/launch{renders}")
        ras_traffic_raw, _ = counted_traffic(f"rasteriser/{pano_w}x{pano_h}/launch{renders}", "bytes_per_unit_raw_fetch")
        ver_traffic, ver_src = counted_traffic(f"verifier/resnet{args.layers}-{6 * S}ch/launch{full_n}")
        ver_alg, _ = counted_traffic(f"verifier_algorithmic/resnet{args.layers}-{6 * S}ch")
        issue = counted_issue(f"densify_issue/{pano_w}x{pano_h}/launch{renders}")
        config5 = (pano_h, pano_w, S, args.layers) == (1024, 2048, 2, 152)
        out = {
            "metric": "alignment hypotheses/sec (render+verify)", "value": round(value, 2), "unit": "hypotheses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "config": {"workload": f"{'BASELINE config 5: ' if config5 else ''}{args.hyps} hypotheses/GPU over {args.panos} synthetic {pano_w}x{pano_h} panoramas "
                                   f"({args.scene} scene), {' + '.join(surfaces)} surface{'s' if S > 1 else ''}, HIP BEV rasteriser + ResNet-{args.layers} "
                                   f"({6 * S}-ch early fusion) fp16 MFMA verifier",
                       "shape": shape, "hypotheses_per_gpu": args.hyps, "panos": args.panos, "renders_per_hypothesis": S,
                       "cached_identity_renders": args.panos * S, "chunk": args.chunk, "hip_streams": 1 if args.no_overlap else args.streams,
                       "parallelism": f"hypothesis-shard x{world}", "rccl": bool(use_dist)},
            # the rasteriser as a whole (splat + densify): SURVEY 8d's bytes per render x the renders of one launch /
            # the summed average durations of those launches (HIP events on the launching streams).  The pose-independent
            # panorama index (bev_pano_index_kernel, once per panorama set at load_panos) is outside the step, like the uploads.
            "roofline": {"kernel": "rasteriser: bev_splat_kernel + bev_densify_kernel", "bound": "hbm",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": None if ras_traffic is None else int(ras_traffic * renders),
                         "traffic_raw_fetch": None if ras_traffic_raw is None else int(ras_traffic_raw * renders), "traffic_source": ras_src,
                         "launch_ms": round(ras_ms, 3), "scatter_ms": round(scat_ms, 3), "densify_ms": round(dens_ms, 3),
                         "launches_timed": len(vfull), "renders_per_launch": renders, "algorithmic_bytes_per_render": bpr,
                         # the bound that BINDS the dominant kernel (bev_densify_kernel): VALU busy share of the SIMD cycles and vector /
                         # scalar wave-instructions per render, from the SQ counter pass at this launch shape (null: no pass at this shape)
                         "valu_busy": None if issue is None else issue.get("valu_busy"),
                         "vector_insts_per_render": None if issue is None else issue.get("vector_insts_per_render"),
                         "scalar_insts_per_render": None if issue is None else issue.get("scalar_insts_per_render"),
                         "issue_source": None if issue is None else issue.get("source")},
            # the verifier against BOTH of its roofs: the dense fp16 MFMA peak (frac) and the HBM time of its activation traffic at
            # the present fusion level (bound_hbm_ms = algorithmic activation + weight bytes / 6.3 TB/s achievable)
            "roofline_verifier": {"kernel": "stem_pool / bottleneck / conv kernels of one ResNet forward", "bound": "mfma",
                                  "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(tflops / MFMA_PEAK_TFLOPS, 5), "launch_ms": round(ver_ms, 3),
                                  "samples_per_launch": full_n, "gflop_per_sample": gflop,
                                  "bound_mfma_ms": round(full_n * gflop / MFMA_PEAK_TFLOPS, 3),
                                  "bound_hbm_ms": None if ver_alg is None else round(ver_alg * full_n / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3, 3),
                                  "traffic": None if ver_traffic is None else int(ver_traffic * full_n), "traffic_source": ver_src},
        }
        if world == 1 and not args.no_power_probe:
            # the verifier forward in a sustained loop (outside the timed region) while a thread reads the GPU's hwmon files: power and the clock it leaves
            probe = power_probe(lambda: pipe.engine.forward_nhwc(pipe.tile_bufs[0][:full_n], out=logits[:full_n]), device_index=local_rank)
            if probe is not None:
                out["roofline_verifier"].update({"power_w": probe["power_w"], "power_cap_w": probe["power_cap_w"], "sclk_mhz": probe["sclk_mhz"], "power_samples": probe["samples"],
                                                 "peak_at_sclk": round(MFMA_PEAK_TFLOPS * probe["sclk_mhz"] / 2400.0, 1),
                                                 "frac_at_sclk": round(tflops / (MFMA_PEAK_TFLOPS * probe["sclk_mhz"] / 2400.0), 5)})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
