"""Development (GPU box): does the ORDER of the renders inside a launch move the densify kernel?  A launch of 4096 renders is eight
rounds of 512 resident workgroups; renders differ in cost, so the launch ends with a tail in which CUs run dry.  Orders tried: the
product's (by panorama, for the splat's L2 locality), by the in-window point count of a first pass descending (longest first) and
ascending, random, counts of sites with missing neighbours descending.  (The probe that led to bev_cost_kernel / bev_order_kernel: the
library now orders the renders itself; the probe sets out_flags bit 4 -- "as given" -- so that the host-side orders decide again, and
prints the library's own order as the last line.)   usage: python tools/probe/densify_order_probe.py [scene]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.pipeline import RenderVerifyPipeline
scene = sys.argv[1] if len(sys.argv) > 1 else "box"
dev = torch.device("cuda:0")
N, P = 4096, 64
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
synthetic.trained_looking_batchnorm(model)
pipe = RenderVerifyPipeline(model, dev, chunk=N, overlap=False, streams=1)
panos = [synthetic.make_pano(i, scene=scene) for i in range(P)]
pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
pipe.ras.cfg.out_flags = 4   # densify in the order given
table = synthetic.make_hypotheses(N, P, seed=0)
prep = pipe.prepare(table)
ref = pipe.score(prep).clone()
torch.cuda.synchronize()
inw = prep["in_window"].cpu().numpy()[prep["rank"]]          # per hypothesis
print("in-window points per render: min %d  median %d  max %d" % (inw.min(), np.median(inw), inw.max()))
# features of a render's SITE set, from the sparse images of a scatter alone (torch on the device, in slices)
S_ = 1
pipe._scatter_chunk(prep, 0, N, 0, 0)
torch.cuda.synchronize()
sites = np.zeros(N); b1 = np.zeros(N); b2 = np.zeros(N); b3 = np.zeros(N)
for lo in range(0, N, 256):
    occ = (pipe.bevs[0][lo:lo + 256] != 0)
    pad = torch.nn.functional.pad(occ, (1, 1, 1, 1))
    nb = pad[:, :-2, 1:-1].int() + pad[:, 2:, 1:-1].int() + pad[:, 1:-1, :-2].int() + pad[:, 1:-1, 2:].int()
    nb8 = nb + pad[:, :-2, :-2].int() + pad[:, :-2, 2:].int() + pad[:, 2:, :-2].int() + pad[:, 2:, 2:].int()
    sites[lo:lo + 256] = occ.flatten(1).sum(1).cpu().numpy()
    b1[lo:lo + 256] = (occ & (nb < 4)).flatten(1).sum(1).cpu().numpy()      # a 4-neighbour missing
    b2[lo:lo + 256] = (occ & (nb <= 2)).flatten(1).sum(1).cpu().numpy()     # two or more missing
    b3[lo:lo + 256] = (occ & (nb8 <= 3)).flatten(1).sum(1).cpu().numpy()    # five or more of the eight missing
# (r6) the number the verdict's split launch would order by: the HARD-SITE count of every render (what is left after the lean walks, phase E1),
# read from the development statistics of a densify pass over the same scatter (stats[6]); a split stage -- launch A = phases B .. E1 + F,
# an order kernel on these counts, launch B = E2 + G -- can be no better than ordering the ONE launch by them, which is what is timed below
import ctypes
from salve_amd import status
stats = torch.zeros((N, 8), dtype=torch.int32, device=dev)
ws = pipe.ras._workspace(N)
scratch = pipe.bevs[0].clone()
st = pipe.ras.lib.salve_bev_densify(ctypes.byref(pipe.ras.cfg), N, ctypes.c_void_p(scratch.data_ptr()), None, ctypes.c_void_p(stats.data_ptr()), status.ptr(dev),
                                    ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
assert st == 0
torch.cuda.synchronize()
hard = stats[:, 6].cpu().numpy().astype(np.float64)
del scratch
# (r6) a PRIOR that needs no split launch: the hard-site count of the render's PANORAMA at identity pose (known once per panorama set, from load_panos'
# identity renders) -- does a panorama's count predict its posed renders'?
from salve_amd.rasteriser import pack_hypotheses, SURFACES
ih = pack_hypotheses(np.arange(P), np.zeros(P, dtype=np.int64), np.tile(np.eye(2, dtype=np.float32), (P, 1, 1)), np.zeros((P, 2), np.float32), np.zeros(P))
_, dbg = pipe.ras.render(pipe.pano_rgb, pipe.pano_depth, pipe.ras.upload_hypotheses(ih), P, debug=True)
torch.cuda.synchronize()
hard_ident = dbg.stats[:, 6].cpu().numpy().astype(np.float64)
prior = hard_ident[np.asarray(table.i1)][np.argsort(prep["rank"])]      # render order, like the other features (converted below)
print("identity hard sites per panorama: min %d median %d max %d; correlation with the posed renders' true count: %.2f" %
      (hard_ident.min(), np.median(hard_ident), hard_ident.max(), np.corrcoef(prior, hard)[0, 1]))
feat = {"HARD sites (true count after E1)": hard, "panorama's identity hard sites (prior)": prior, "sites": sites, "sites with a 4-neighbour missing": b1, "sites with >= 2 4-neighbours missing": b2, "sites with >= 5 of 8 neighbours missing": b3}
feat = {k: v[prep["rank"]] for k, v in feat.items()}   # render order -> hypothesis order
for k, v in feat.items():
    print(f"{k}: min {v.min():.0f} median {np.median(v):.0f} max {v.max():.0f}")
rng = np.random.default_rng(0)
orders = {"by panorama (product)": None, "in-window descending": np.argsort(-inw, kind="stable"), "random": rng.permutation(N)}
for k, v in feat.items():
    orders[k + ", descending"] = np.argsort(-v, kind="stable")
for wgt in (0.01, 0.03):
    orders[f"hard + {wgt} sites, descending"] = np.argsort(-(feat["HARD sites (true count after E1)"] + wgt * feat["sites"]), kind="stable")
for wgt in (2.0, 5.0, 10.0):
    orders[f"b2 + {wgt} x panorama prior, descending"] = np.argsort(-(feat["sites with >= 2 4-neighbours missing"] + wgt * feat["panorama's identity hard sites (prior)"]), kind="stable")
orders["b2 + 0.05 sites, descending"] = np.argsort(-(feat["sites with >= 2 4-neighbours missing"] + 0.05 * feat["sites"]), kind="stable")
for wgt in (2, 4, 8):
    orders[f"b1 + {wgt} b3, descending"] = np.argsort(-(feat["sites with a 4-neighbour missing"] + wgt * feat["sites with >= 5 of 8 neighbours missing"]), kind="stable")
srt = np.argsort(-feat["sites with >= 2 4-neighbours missing"], kind="stable")
for mix in (128, 256):
    head = np.empty(2 * mix, dtype=np.int64)
    head[0::2], head[1::2] = srt[:mix], srt[::-1][:mix]
    orders[f"b2 descending, the first {2 * mix} alternating with the {mix} cheapest"] = np.concatenate([head, srt[mix:N - mix]])
for mix in (256,):   # the first round: the costliest alternating with MEDIAN renders (phases out of step from the start; the cheapest stay for the tail)
    mid = srt[N // 2: N // 2 + mix]
    head = np.empty(2 * mix, dtype=np.int64)
    head[0::2], head[1::2] = srt[:mix], mid
    rest = np.concatenate([srt[mix:N // 2], srt[N // 2 + mix:]])
    orders[f"b2 descending, the first {2 * mix} alternating with {mix} median ones"] = np.concatenate([head, rest])
orders["b3 + 0.02 sites, descending"] = np.argsort(-(feat["sites with >= 5 of 8 neighbours missing"] + 0.02 * feat["sites"]), kind="stable")
res = {}
for name, order in orders.items():
    prep2 = pipe.prepare(table, order=order)
    for _ in range(2):
        out = pipe.score(prep2)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), name
    ev = []
    for _ in range(6):
        pipe.score(prep2, timers=ev)
    torch.cuda.synchronize()
    ms = lambda tag: float(np.mean([a.elapsed_time(b) for a, b, r, t in ev if t == tag]))
    print(f"{name:55s} scatter {ms('scatter'):.3f} ms   densify {ms('densify'):.3f} ms", flush=True)
pipe.ras.cfg.out_flags = 0
prep2 = pipe.prepare(table)
ev = []
for _ in range(8):
    out = pipe.score(prep2, timers=ev if _ >= 2 else None)
torch.cuda.synchronize()
assert torch.equal(out, ref)
print(f"{'the library: by panorama + costly first (product)':55s} scatter {ms('scatter'):.3f} ms   densify {ms('densify'):.3f} ms", flush=True)
