import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from types import SimpleNamespace
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.pipeline import RenderVerifyPipeline
dev = torch.device("cuda:0")
P, N = 8, 96
panos = [synthetic.make_pano(i) for i in range(P)]
torch.manual_seed(0)
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
table = synthetic.make_hypotheses(N, P, seed=1)
res = {}
for overlap in (False, True):
    pipe = RenderVerifyPipeline(model, dev, chunk=64, overlap=overlap)
    pipe.load_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    prep = pipe.prepare(table)
    outs = []
    for rep in range(3):
        o = pipe.score(prep); torch.cuda.synchronize()
        outs.append((o.clone(), [b.clone() for b in pipe.bevs], [t.clone() for t in pipe.tile_bufs]))
    res[overlap] = outs
base = res[False][0]
for overlap in (False, True):
    for rep in range(3):
        o, bevs, tiles = res[overlap][rep]
        d = (o != base[0]).any(1).nonzero().flatten().tolist()
        print("overlap", overlap, "rep", rep, "logit rows differing vs base:", d[:10], len(d))
        if overlap:
            # chunk0 -> buf0 (64), chunk1 -> buf1 (32)
            b0 = (bevs[0] != base[1][0]).reshape(64, -1).any(1).nonzero().flatten().tolist()
            print("   bev buf0 renders differing from non-overlap chunk... (non-overlap buf0 holds chunk1):", len(b0))
            t0 = (tiles[0] != res[True][0][2][0]).reshape(64, -1).any(1).nonzero().flatten().tolist()
            t1 = (tiles[1] != res[True][0][2][1]).reshape(64, -1).any(1).nonzero().flatten().tolist()
            print("   tiles vs overlap rep0: buf0 diff", t0[:8], "buf1 diff", t1[:8])
print("---- pixel-level")
ref_bev1 = res[False][0]  # non-overlap: buf0 holds last chunk (chunk1, 32 renders)
for rep in range(3):
    bevs = res[True][rep][1]
    d = (bevs[1][:32] != ref_bev1[1][0][:32])
    idx = d.reshape(32, -1).any(1).nonzero().flatten().tolist()
    print("rep", rep, "renders of chunk1 differing from non-overlap:", idx)
    for r in idx[:3]:
        ys, xs = d[r].nonzero(as_tuple=True)
        print("   render", r, "n px", len(ys), "y range", int(ys.min()), int(ys.max()), "x range", int(xs.min()), int(xs.max()),
              "vals ovl", bevs[1][r][ys[:4], xs[:4]].tolist(), "ref", ref_bev1[1][0][r][ys[:4], xs[:4]].tolist())
