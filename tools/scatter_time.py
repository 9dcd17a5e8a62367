import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
n = 512
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i) for i in range(4)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, 4, seed=0)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[:n], np.zeros(n), hyp.R[:n], hyp.t[:n], np.ones(n)))
for flags in (0, 0):
    ras.cfg.reserved1 = flags
    for _ in range(2): ras.scatter(d_rgb, d_depth, hd, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ras.scatter(d_rgb, d_depth, hd, n)
    torch.cuda.synchronize()
    print("flags", flags, "scatter+memset per render us:", (time.perf_counter() - t0) / 5 / n * 1e6)
