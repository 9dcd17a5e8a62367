"""Where the densify kernel's time goes, on the product library (GPU box): the kernel is timed whole and with one phase
switched off at a time through the development flags of the config (reserved1: 1 = no stars at all, 4 = no general walk,
8 = no triangle rasterisation).  The images of the ablated runs are wrong, of course; only the times are used."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scene = sys.argv[2] if len(sys.argv) > 2 else "box"
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i, scene=scene) for i in range(8)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, 8, seed=0)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[:n], np.zeros(n), hyp.R[:n], hyp.t[:n], np.ones(n)))
def timed(flags, reps=3):
    ras.cfg.reserved1 = flags
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ras.render(d_rgb, d_depth, hd, n)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e6 / n
full = timed(0)
print(f"{scene}: scatter + densify, {n} renders per launch: {full:.2f} us per render")
for name, fl in (("E2 walks but does not rasterise", 2), ("without the general walk (E2)", 4), ("without triangle rasterisation (F)", 8), ("without E2 and F", 12), ("without any star walk (E1, E2, F)", 1)):
    t = timed(fl)
    print(f"  {name:40s} {t:.2f} us   (-{full - t:.2f})")
