"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (used by __graft_entry__.smoke)."""

from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


def run_smoke() -> None:
    from oracle import bev_oracle as bo  # the checker, never the thing shipped
    from oracle import resnet_oracle as ro
    from salve_amd import synthetic
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    from salve_amd.pipeline import RenderVerifyPipeline

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    panos = [synthetic.make_pano(i) for i in range(2)]
    hyp = synthetic.make_hypotheses(2, 2, seed=0)
    pipe = RenderVerifyPipeline(model, dev, chunk=2)
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    prepared = pipe.prepare(hyp)
    logits = pipe.score(prepared)
    torch.cuda.synchronize()
    bev_gpu = pipe.ras.export_u8(pipe.bev[:2]).cpu().numpy()
    tiles_gpu = pipe.tiles[:2].float().cpu()

    # oracle: render both panos of hypothesis 0 and compare the BEV image and the logits
    a = bo.xyzrgb_from_arrays(panos[hyp.i1[0]][1], panos[hyp.i1[0]][0], bo.floor_ceiling_z_range("floor"))
    b = bo.xyzrgb_from_arrays(panos[hyp.i2[0]][1], panos[hyp.i2[0]][0], bo.floor_ceiling_z_range("floor"))
    a, b = bo.pose_pair(a, b, hyp.R[0], hyp.t[0])
    r1 = bo.render_bev_image(a, mode="exact")
    r2 = bo.render_bev_image(b, mode="exact")
    assert np.array_equal(bev_gpu[0], r1["bev"]), "BEV image differs from the oracle"
    t = torch.from_numpy(np.concatenate([bo.tile_from_bev(r1["bev"]), bo.tile_from_bev(r2["bev"])], 0))
    assert torch.equal(tiles_gpu[0, :, :, :6].permute(2, 0, 1), t.bfloat16().float()), "tiles differ from the oracle"
    with torch.no_grad():
        ref = ro.forward(model.state_dict(), 18, [t[None, :3].bfloat16().float(), t[None, 3:].bfloat16().float()])
    err = float((logits[:1].cpu() - ref).abs().max())
    assert err < 3e-2 * max(1.0, float(ref.abs().max())), f"logit error {err}"
    print(f"smoke OK: BEV bit-exact, logits {logits[0].tolist()} (oracle {ref[0].tolist()}, err {err:.2e})")
