import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
# usage: pmc_render.py [renders per launch = 4096 (the benchmark's launch shape)] [panoramas = 64]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
ras.cfg.reserved1 = int(os.environ.get("SALVE_DBG_FLAGS", "0"))
panos = [synthetic.make_pano(i) for i in range(P)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, P, seed=0)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[:n], np.zeros(n), hyp.R[:n], hyp.t[:n], np.ones(n)))
for _ in range(2):
    ras.render(d_rgb, d_depth, hd, n)
torch.cuda.synchronize()
