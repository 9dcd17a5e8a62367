import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
# usage: pmc_render.py [renders per launch = 4096 (the benchmark's launch shape)] [panoramas = 64] [HxW = 512x1024] [surfaces = floor | floor,ceiling]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "512x1024").split("x"))
surfs = (sys.argv[4] if len(sys.argv) > 4 else "floor").split(",")
dev = torch.device("cuda:0")
ras = BevRasteriser(dev, pano_hw=(H, W))
ras.cfg.reserved1 = int(os.environ.get("SALVE_DBG_FLAGS", "0"))
panos = [synthetic.make_pano(i, H, W) for i in range(P)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
S = len(surfs)
hyp = synthetic.make_hypotheses(n // S, P, seed=0)          # n renders = n / S hypotheses x S surfaces, as the pipeline issues them
order = np.argsort(hyp.i1, kind="stable")                  # ... in panorama order (pipeline.prepare)
sid = [0 if v == "floor" else 1 for v in surfs]
hd = ras.upload_hypotheses(pack_hypotheses(np.repeat(hyp.i1[order], S), np.tile(sid, n // S), np.repeat(hyp.R[order], S, axis=0),
                                           np.repeat(hyp.t[order], S, axis=0), np.ones(n)))
# The rasteriser as the pipeline launches it since round 6: the scatter stage, then the densify stage whose last phase writes every render's
# verifier tile (salve_bev_densify_tiles): render r -> sample r // S, channel group r % S; the pair's second image = a pretiled image per panorama.
crop = ras.crop
in_c = 8 * S
bev = torch.empty((n, *ras.bev_hw), dtype=torch.int32, device=dev)
tiles_b = torch.zeros((P * S, crop, crop), dtype=torch.int32, device=dev)
tiles = torch.zeros((n // S, crop, crop, in_c), dtype=torch.float16, device=dev)
r = np.arange(n)
jobs_a = ras.upload_tile_jobs(np.zeros(n, dtype=np.int64), r // S, 6 * (r % S))
jobs_b = ras.upload_tile_jobs(np.repeat(hyp.i2[order], S) * S + (r % S), r // S, 6 * (r % S) + 3, pretiled=True)
for _ in range(2):
    ras.scatter(d_rgb, d_depth, hd, n, bev)
    ras.densify_tiles(n, bev, jobs_a, jobs_b, tiles_b, tiles, in_c)
torch.cuda.synchronize()
ras.check("pmc_render")
