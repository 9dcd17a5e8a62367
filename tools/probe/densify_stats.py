"""Per-render work counters of the densify kernel (product build, debug outputs on): sites, listed sites, lean-walk lane
iterations, hard sites, queued triangles (GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
n = 64
scene = sys.argv[1] if len(sys.argv) > 1 else "box"
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i, scene=scene) for i in range(4)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, 4, seed=0)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[:n], np.zeros(n), hyp.R[:n], hyp.t[:n], np.ones(n)))
bev, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
torch.cuda.synchronize()
st = dbg.stats.cpu().numpy().astype(np.float64)
names = ["sites", "sites on the list", "(checksum)", "rows", "lean lane-iterations", "err", "hard sites", "queued general triangles"]
print(scene, {k: round(float(v), 1) for k, v in zip(names, st.mean(0))})
print("lean iterations per listed site:", round(float(st[:, 4].sum() / st[:, 1].sum()), 2))
