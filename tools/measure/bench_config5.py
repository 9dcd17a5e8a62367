"""BASELINE config 5 on one GPU (GPU box): 2048 x 1024 synthetic panoramas, floor + ceiling, ResNet-152 with 12 input channels,
fp16 -- hypotheses per second of the fused render -> verify pipeline, with the per-stage HIP-event times.  Not the benchmark
line (bench.py measures config 2/3); a record of where the larger configuration stands.
usage: python tools/measure/bench_config5.py [hypotheses=1024] [panos=16] [chunk=512]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.pipeline import RenderVerifyPipeline

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 512
H, W = 1024, 2048
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = EarlyFusionCEResnet(152, False, 2, SimpleNamespace(modalities=["ceiling_rgb_texture", "floor_rgb_texture"])).eval()
synthetic.trained_looking_batchnorm(model, seed=0)
pipe = RenderVerifyPipeline(model, dev, pano_hw=(H, W), chunk=chunk, overlap=False, streams=1)
panos = [synthetic.make_pano(i, H, W) for i in range(P)]
pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(N, P, seed=0)
prep = pipe.prepare(hyp)
pipe.score(prep); torch.cuda.synchronize()
timers, vt = [], []
t0 = time.perf_counter()
logits = pipe.score(prep, timers=timers, vtimers=vt)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pipe.check()
sc = sum(a.elapsed_time(b) for a, b, u, tag in timers if tag == "scatter")
de = sum(a.elapsed_time(b) for a, b, u, tag in timers if tag == "densify")
ve = sum(a.elapsed_time(b) for a, b, n in vt)
print(f"config 5 on one GPU: {N} hypotheses ({2 * N} renders of {W}x{H} panoramas, ResNet-152 12-ch), chunk {chunk}: {N / dt:.0f} hypotheses/s")
print(f"  scatter {sc:.1f} ms ({sc * 1e3 / (2 * N):.2f} us per render), densify {de:.1f} ms ({de * 1e3 / (2 * N):.2f} us per render), "
      f"verifier {ve:.1f} ms ({N * 23.731 / ve:.0f} TFLOP/s), everything {dt * 1e3:.1f} ms")
