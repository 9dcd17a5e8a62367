"""Benchmark of the SALVe hot path: alignment hypotheses / second (render + verify) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU under torch.distributed.run (RCCL).  When the script is started WITHOUT a launcher
(`python bench.py --gpus 8`), it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a
child process -- before anything touches the GPU -- and relays the child's JSON line and exit code.

A step = one pass of the fused render+verify path over the rank's shard of the hypothesis table
(BASELINE.json configs[2]: 4096 hypotheses over 64 synthetic 1024x512 panoramas, rasteriser + ResNet-50 fp16,
per GPU -> weak scaling).  Inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

At N = 1 the same JSON line also carries, all measured OUTSIDE the timed region:
  * `roofline.hbm_copy_gbs_measured` -- a 4 GB device copy on this box ("confirm with a copy microbenchmark on the box");
  * `roofline_verifier.gemm_*` -- a plain fp16 GEMM (fp32 accumulation, random data, torch.matmul = hipBLASLt: a MEASUREMENT REFERENCE,
    never product) at the three 3 x 3 shapes of the forward + one square shape, each with board power and shader clock: what the matrix
    pipes of THIS box sustain at its power cap, beside what conv8_kernel gets;
  * `config5` -- BASELINE.json configs[4] (2048 x 1024 panoramas, floor + ceiling, ResNet-152 12-channel), 3 timed steps at the launch
    size the pipeline picks (pipeline.pick_launch).
BASELINE.json configs[4] alone (same JSON schema, `config.workload` names it):
    python bench.py --pano-hw 1024x2048 --surfaces floor,ceiling --layers 152 --panos 16
`--chunk N` fixes the hypotheses per launch (default: chosen by the pipeline from the free HBM -- the whole shard when it fits).
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


def power_probe(fn, seconds: float = 3.0, device_index: int = 0, batch: int = 4, want_rate: bool = False):
    """Board power and shader clock while `fn` (one verifier forward) runs in a sustained loop, OUTSIDE the timed region: the forward runs
    at the board's power limit, so the clock the matrix pipes actually get -- not the 2.4 GHz the 2.5 PFLOP/s peak
    assumes -- is part of what the roofline fraction means.  Read from the GPU's hwmon files in sysfs by a thread while the loop runs (no
    child process); None where sysfs does not show the device."""
    import threading

    hw = _hwmon_of(device_index)
    if hw is None and not want_rate:
        return None
    samples, stop = [], threading.Event()

    def sampler():
        if hw is None:
            return
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
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    calls, gpu_ms = 0, 0.0
    while time.perf_counter() - t0 < seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(batch):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 1.0 or seconds <= 1.0:   # rate: only once the clocks have settled under load
            calls += batch
            gpu_ms += e0.elapsed_time(e1)
    stop.set()
    th.join(timeout=5)
    rate = {"ms_per_call": (gpu_ms / calls) if calls else None, "calls_timed": calls}
    if not samples:
        return rate if want_rate else None
    try:
        cap = int(open(hw + "/power1_cap").read()) / 1e6
    except Exception:
        cap = None
    return {"power_w": round(float(np.mean([a for a, _ in samples])), 1), "sclk_mhz": round(float(np.mean([b for _, b in samples])), 1),
            "power_cap_w": cap, "samples": len(samples), **rate}


# The three 3 x 3 convolutions of the ResNet-50 forward as GEMMs (reference shapes: salve/models/resnet_factory.py:26-44 = torchvision's
# Bottleneck widths; batch 4096 x output pixels, C_out, 9 C_in) and one square shape whose operands' HBM time is far below its matrix time.
GEMM_SHAPES = [("l3_3x3_256to256_at14", 4096 * 196, 256, 2304), ("l2_3x3_128to128_at28", 4096 * 784, 128, 1152),
               ("l4_3x3_512to512_at7", 4096 * 49, 512, 4608), ("square_8192", 8192, 8192, 8192)]


def gemm_probe(dev, device_index: int, seconds: float = 2.5):
    """What a plain fp16 GEMM with fp32 accumulation sustains on THIS box at its power cap, on random data (the power an MFMA draws
    depends on the bits it multiplies): torch.matmul (hipBLASLt) -- a measurement reference beside the product's convolution kernels,
    never part of the product path.  A[M, K] x W[N, K]^T, the layout of an im2col'd activation against packed weights.  An explicit
    A matrix is nine times the bytes the implicit GEMM reads, so the shapes with small N K / (N + K) are HBM-bound AS GEMMs
    (`flop_per_byte` x the copy rate is their cap); the square shape and the N = 512 shape are not."""
    out = {}
    for name, M, N, K in GEMM_SHAPES:
        try:
            a = torch.randn((M, K), dtype=torch.float16, device=dev)
            w = torch.randn((N, K), dtype=torch.float16, device=dev)
            c = torch.empty((M, N), dtype=torch.float16, device=dev)
            fn = lambda: torch.matmul(a, w.t(), out=c)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            r = power_probe(fn, seconds=seconds, device_index=device_index, batch=8, want_rate=True)
            flop = 2.0 * M * N * K
            e = {"m": M, "n": N, "k": K, "tflops": None if not r or not r.get("ms_per_call") else round(flop / r["ms_per_call"] / 1e9, 1),
                 "flop_per_byte": round(flop / (2.0 * (M * K + N * K + M * N)), 1)}
            if r:
                e.update({k: r[k] for k in ("power_w", "sclk_mhz") if k in r})
            out[name] = e
            del a, w, c
        except Exception as ex:   # a box whose free memory or BLAS build cannot run a shape reports that, the benchmark line stays valid
            out[name] = {"m": M, "n": N, "k": K, "tflops": None, "error": str(ex)[:120]}
        torch.cuda.empty_cache()
    return out


COPY_WGS_PER_CU, COPY_MODE = 64, 1   # launch shape of the copy kernel: the best of tools/measure/copy_sweep.py (non-temporal, 4 loads in flight, 64 workgroups per CU: 5.3 TB/s; torch copy_ 4.9)


def copy_probe(dev, gbytes: float = 4.0, iters: int = 20):
    """HBM copy rate of this box: a float4 device-to-device copy of 4 GB, bytes read + bytes written per second.  The kernel is the
    measurement helper of tests/native (a grid-stride float4 copy, 8 workgroups per CU; not part of the product library); where that
    library is not built, torch's `copy_` (the runtime's blit) is timed instead and the line says so.  Returns (GB/s, what ran)."""
    import ctypes

    n = int(gbytes * (1 << 30)) // 16
    src = torch.empty((n, 4), dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    fn, what = (lambda: dst.copy_(src)), "torch copy_ (runtime blit)"
    helper = ROOT / "tests" / "native" / "libsalve_testhelp.so"
    if helper.exists():
        try:
            lib = ctypes.CDLL(str(helper))
            lib.salve_debug_copy16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
            lib.salve_debug_copy16.restype = ctypes.c_int
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            stream = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

            def fn():
                if lib.salve_debug_copy16(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), n, COPY_WGS_PER_CU * cus, COPY_MODE, stream()) != 0:
                    raise RuntimeError("salve_debug_copy16 failed")
            what = "float4 grid-stride copy kernel (tests/native/testhelp.hip)"
        except (OSError, AttributeError):
            pass
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ok = bool(torch.equal(dst[:1024], src[:1024]) and torch.equal(dst[-1024:], src[-1024:]))
    del src, dst
    torch.cuda.empty_cache()
    return (round(2.0 * n * 16 / (ms * 1e-3) / 1e9, 1) if ok else None), what


def make_panos(n: int, pano_h: int, pano_w: int, scene: str):
    """The synthetic panoramas of a run (salve_amd.synthetic.make_panos: a few host threads)."""
    from salve_amd import synthetic

    return synthetic.make_panos(n, pano_h, pano_w, scene=scene)


def config5_line(dev, hyps: int = 4096, panos: int = 16, steps: int = 3, warmup: int = 1):
    """BASELINE.json configs[4] on this GPU, after the main region: 2048 x 1024 panoramas, floor + ceiling, ResNet-152 with the 12-channel
    early fusion (the reference's released models: salve/configs/*.yaml, num_layers 152), `hyps` hypotheses at the launch size the pipeline
    picks, `steps` timed passes between device synchronisations."""
    from salve_amd import synthetic
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    H, W, S = 1024, 2048, 2
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(152, False, 2, SimpleNamespace(modalities=["ceiling_rgb_texture", "floor_rgb_texture"])).eval()
    synthetic.trained_looking_batchnorm(model)
    pipe = RenderVerifyPipeline(model, dev, pano_hw=(H, W), chunk=None, overlap=False, streams=1, n_hypotheses=hyps)
    ps = make_panos(panos, H, W, "box")
    pipe.load_panos(np.stack([p[0] for p in ps]), np.stack([p[1] for p in ps]))
    prepared = pipe.prepare(synthetic.make_hypotheses(hyps, panos, seed=0))
    logits = torch.empty((hyps, 2), dtype=torch.float32, device=dev)
    for _ in range(warmup):
        pipe.score(prepared, out=logits)
    ev, vev = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.score(prepared, out=logits, timers=ev, vtimers=vev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.check("bench.py config 5")
    assert bool(torch.isfinite(logits).all())
    launch = min(pipe.chunk, hyps)
    renders = launch * S
    mean_ms = lambda tag: float(np.mean([a.elapsed_time(b) for a, b, r, t in ev if t == tag and r == renders]))
    scat_ms, dens_ms = mean_ms("scatter"), mean_ms("densify")
    ver_ms = float(np.mean([a.elapsed_time(b) for a, b, r in vev if r == launch]))
    gflop = GFLOP_PER_SAMPLE[(152, 12)]
    achieved = renders * bytes_per_render(H, W) / ((scat_ms + dens_ms) * 1e-3) / 1e9
    tflops = launch * gflop / ver_ms
    return {"workload": f"BASELINE config 5: {hyps} hypotheses over {panos} synthetic {W}x{H} panoramas (box scene), floor + ceiling, HIP BEV rasteriser + "
                        f"ResNet-152 (12-ch early fusion) fp16 MFMA verifier", "value": round(hyps * steps / dt, 2), "unit": "hypotheses/s",
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "launch": launch, "renders_per_launch": renders,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "scatter_ms": round(scat_ms, 3), "densify_ms": round(dens_ms, 3), "algorithmic_bytes_per_render": bytes_per_render(H, W)},
            "roofline_verifier": {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(tflops / MFMA_PEAK_TFLOPS, 5), "launch_ms": round(ver_ms, 3), "gflop_per_sample": gflop}}


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
    ap.add_argument("--chunk", type=int, default=0, help="hypotheses per render / verify launch; 0 (default): the pipeline picks the largest launch that\n"
                    "fits half of the free HBM -- the whole shard here (launches of 4096 run 10 %% faster per unit than launches of 1024)")
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
    ap.add_argument("--no-calibration", action="store_true", help="skip the GEMM and copy microbenchmarks (measured roofs of this box)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE config 5 sub-measurement")
    args = ap.parse_args()

    # before anything starts a rank or touches a GPU: are there that many devices?  (device_count() does not initialise the GPU)
    have = torch.cuda.device_count()
    if args.gpus < 1 or args.gpus > have:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); nothing was launched")

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
    pipe = RenderVerifyPipeline(model, dev, pano_hw=(pano_h, pano_w), chunk=(args.chunk or None), overlap=(args.streams > 1 and not args.no_overlap),
                                streams=args.streams, n_hypotheses=args.hyps)
    panos = make_panos(args.panos, pano_h, pano_w, args.scene)
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
        full_n = min(pipe.chunk, len(table))           # hypotheses per launch
        renders = full_n * S                           # renders per rasteriser launch
        bpr = bytes_per_render(pano_h, pano_w)
        mean_ms = lambda tag: float(np.mean([a.elapsed_time(b) for a, b, r, t in ev if t == tag and r == renders]))
        scat_ms, dens_ms = mean_ms("scatter"), mean_ms("densify")
        ras_ms = scat_ms + dens_ms
        achieved = renders * bpr / (ras_ms * 1e-3) / 1e9
        tile_bytes = 224 * 224 * (2 * (pipe.engine.in_channels // S) + 4)   # per render: its share of the fp16 NHWC sample + the u8x4 second image
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
                       "cached_identity_renders": args.panos * S, "chunk": pipe.chunk, "chunk_chosen_by": "--chunk" if args.chunk else "pipeline.pick_launch", "hip_streams": 1 if args.no_overlap else args.streams,
                       "parallelism": f"hypothesis-shard x{world}", "rccl": bool(use_dist)},
            # the rasteriser as a whole (splat + densify): SURVEY 8d's bytes per render x the renders of one launch /
            # the summed average durations of those launches (HIP events on the launching streams).  The pose-independent
            # panorama index (bev_pano_index_kernel, once per panorama set at load_panos) is outside the step, like the uploads.
            "roofline": {"kernel": "rasteriser: bev_splat_kernel + bev_densify_kernel (whose last phase writes the verifier tiles since round 6: salve_bev_densify_tiles)", "bound": "hbm",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": None if ras_traffic is None else int(ras_traffic * renders),
                         "traffic_raw_fetch": None if ras_traffic_raw is None else int(ras_traffic_raw * renders), "traffic_source": ras_src,
                         "launch_ms": round(ras_ms, 3), "scatter_ms": round(scat_ms, 3), "densify_ms": round(dens_ms, 3),
                         "launches_timed": len(vfull), "renders_per_launch": renders, "algorithmic_bytes_per_render": bpr,
                         # the tile phase rides in densify_ms; its own bytes (fp16 NHWC sample written, pretiled second image read) are NOT in
                         # SURVEY 8d's figure and not in `achieved`: with them the same launches move this many GB/s
                         "tile_bytes_per_render": tile_bytes, "achieved_with_tile_bytes": round(renders * (bpr + tile_bytes) / (ras_ms * 1e-3) / 1e9, 3),
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
        if world == 1 and not args.no_calibration:
            # the densify stage WITHOUT its tile phase (salve_bev_densify instead of salve_bev_densify_tiles), outside the timed region: since round 6 the
            # verifier tiles are written by the densify kernel's last phase, so `densify_ms` above carries work SURVEY 8d's bytes do not count --
            # this is the stage as rounds 1-5 timed it, for a like-for-like `frac`
            try:
                S_ = len(pipe.surfaces)
                evs = []
                for _ in range(4):
                    pipe._scatter_chunk(prepared, 0, full_n, 0, 0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    pipe.ras.densify(full_n * S_, pipe.bevs[0])
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                d_only = float(np.mean([a.elapsed_time(b) for a, b in evs[1:]]))
                out["roofline"].update({"densify_without_tile_phase_ms": round(d_only, 3),
                                        "frac_without_tile_phase": round(renders * bpr / ((scat_ms + d_only) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)})
            except Exception as ex:
                out["roofline"]["densify_without_tile_phase_ms"] = None
                out["roofline"]["densify_without_tile_phase_error"] = str(ex)[:200]
        if world == 1 and not (args.no_calibration and args.no_config5):
            # everything below needs the memory, not the pipeline: release the main workload's buffers
            del pipe, prepared, logits, allg, table
            torch.cuda.empty_cache()
        # (the extras below run OUTSIDE the timed region; none of them may cost the line its main measurement: a failure is reported in its field)
        if world == 1 and not args.no_calibration:
            try:
                gbs, what = copy_probe(dev)
                out["roofline"].update({"hbm_copy_gbs_measured": gbs, "hbm_copy_kernel": what,
                                        "frac_of_measured_copy": None if not gbs else round(achieved / gbs, 6)})
            except Exception as ex:
                out["roofline"]["hbm_copy_gbs_measured"] = None
                out["roofline"]["hbm_copy_error"] = str(ex)[:200]
            try:
                g = gemm_probe(dev, local_rank)
                out["roofline_verifier"]["gemm_reference"] = g
                conv_shapes = [v["tflops"] for k, v in g.items() if k != "square_8192" and v.get("tflops")]
                best = max([v["tflops"] for v in g.values() if v.get("tflops")], default=None)
                out["roofline_verifier"].update({"gemm_tflops_measured": best, "gemm_tflops_conv_shapes": conv_shapes,
                                                 "gemm_power_w": next((v.get("power_w") for v in g.values() if v.get("tflops") == best), None),
                                                 "gemm_sclk_mhz": next((v.get("sclk_mhz") for v in g.values() if v.get("tflops") == best), None),
                                                 "frac_of_measured_gemm": None if not best else round(tflops / best, 5)})
            except Exception as ex:
                out["roofline_verifier"]["gemm_tflops_measured"] = None
                out["roofline_verifier"]["gemm_error"] = str(ex)[:200]
        if world == 1 and not args.no_config5 and not config5 and args.scene == "box":
            try:
                out["config5"] = config5_line(dev)
            except Exception as ex:
                out["config5"] = {"value": None, "error": str(ex)[:300]}
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as ex:
                out["cpu_baseline"] = {"value": None, "unit": "hypotheses/s", "cores": _cores(), "kind": "port", "sample": "failed", "error": str(ex)[:300]}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
