"""One-off (GPU box): the verifier's logits against the fp32 oracle over several hundred REAL tiles -- rendered by the pipeline from random
hypotheses, so mostly black with textured regions, the input the verifier actually sees -- at realistic logit magnitudes (trained-looking
BatchNorm statistics, classifier scaled by `synthetic.trained_looking_head`).  The oracle is fed the very tiles the GPU verifier read (the
fp16 values, widened), so the difference is the verifier's alone.   usage: python tools/measure/logit_sweep.py [n = 512]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import numpy as np, torch
from oracle import resnet_oracle as ro
from salve_amd import synthetic
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.pipeline import RenderVerifyPipeline
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
for layers, mods, hw, P, scale in ((50, ["floor_rgb_texture"], (512, 1024), 16, 30.0), (152, ["ceiling_rgb_texture", "floor_rgb_texture"], (512, 1024), 16, 30.0)):
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=mods)).eval()
    synthetic.trained_looking_batchnorm(model)
    synthetic.trained_looking_head(model, scale)
    pipe = RenderVerifyPipeline(model, dev, pano_hw=hw, chunk=N, overlap=False, streams=1)
    panos = [synthetic.make_pano(i, *hw, scene="cluttered") for i in range(P)]
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    table = synthetic.make_hypotheses(N, P, seed=5)
    prep = pipe.prepare(table)
    got = pipe.score(prep).cpu()
    pipe.check("logit sweep")
    tiles = pipe.tile_bufs[0][:N].float().cpu().permute(0, 3, 1, 2).contiguous()      # [N, 8 or 16, 224, 224]
    n_img = 2 * len(mods)
    sd = model.state_dict()
    torch.set_num_threads(16)
    t0 = time.time()
    ref = []
    with torch.no_grad():
        for lo in range(0, N, 32):
            x = tiles[lo:lo + 32]
            ref.append(ro.forward(sd, layers, [x[:, 3 * k:3 * k + 3] for k in range(n_img)]))
    ref = torch.cat(ref)
    err = (got - ref).abs().max(1).values
    mag = ref.abs().max(1).values
    perr = (torch.softmax(got, 1) - torch.softmax(ref, 1)).abs().max(1).values
    rel = err / mag.clamp(min=1.0)
    print(f"ResNet-{layers}, {n_img} images, {N} rendered tile sets (oracle {time.time() - t0:.0f} s): |logit| median {float(mag.median()):.2f} max {float(mag.max()):.2f}; "
          f"abs err median {float(err.median()):.2e} p99 {float(err.quantile(0.99)):.2e} max {float(err.max()):.2e}; "
          f"err / max(1, |logit|) p99 {float(rel.quantile(0.99)):.2e} max {float(rel.max()):.2e}; softmax err max {float(perr.max()):.2e}; "
          f"arg-max equal {int((got.argmax(1) == ref.argmax(1)).sum())} / {N}", flush=True)
    del pipe
    torch.cuda.empty_cache()
