"""Stand-alone timing of the rasteriser by phase (GPU box only): renders n BEV images with debug flags
0 (all), 4 (skip hard-site walk), 12 (skip hard sites + rasterisation), 1 (skip stars)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i) for i in range(4)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(max(n, 16), 4, seed=0)
h = pack_hypotheses(hyp.i1[:n], np.zeros(n), hyp.R[:n], hyp.t[:n], np.ones(n))
hd = ras.upload_hypotheses(h)
bev, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
torch.cuda.synchronize()
st = dbg.stats.cpu().numpy()
print("mean stats [sites, begun, -, rows, iters, err, hard, queued]:", st.mean(0).round(0).tolist())
res = {}
for rnd in range(3):
    for flags in (0, 4, 12, 1, 2, 6):
        ras.cfg.reserved1 = flags
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ras.render(d_rgb, d_depth, hd, n)
        torch.cuda.synchronize()
        res.setdefault(flags, []).append((time.perf_counter() - t0) / 3 / n * 1e3)
ras.cfg.reserved1 = 0
print("ms per render (min of 3 rounds):", {k: round(min(v), 5) for k, v in res.items()})
a, b, c, d = (min(res[k]) for k in (0, 4, 12, 1))
print(f"  hard-site walk without its rasterisation: {(min(res[2]) - min(res[6]))*1e3:.1f} us")
print(f"  scatter+bitmaps+mask+base {d*1e3:.1f} us | local walk {(c-d)*1e3:.1f} us | raster {(b-c)*1e3:.1f} us | hard sites {(a-b)*1e3:.1f} us | total {a*1e3:.1f} us")
