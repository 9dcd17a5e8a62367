"""Development (GPU box): a small, low-priority FILLER chunk beside the main one.  The rasteriser of 4096 hypotheses as two launches sets on
two HIP streams -- 4096 - F hypotheses at high priority, F at low priority, each scatter -> densify -> tiles -- so that the filler's
workgroups take the slots the main launches leave free (the densify tail, the tile kernel's spare wave slots), then ONE verifier forward
over the joint tile buffer.  Against the product schedule (one stream, whole-shard launches).   usage: filler_chunk_probe.py [scene]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.synthetic import HypothesisTable
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.pipeline import RenderVerifyPipeline
scene = sys.argv[1] if len(sys.argv) > 1 else "box"
dev = torch.device("cuda:0")
N, P = 4096, 64
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
synthetic.trained_looking_batchnorm(model)
panos = [synthetic.make_pano(i, scene=scene) for i in range(P)]
rgb, depth = np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos])
table = synthetic.make_hypotheses(N, P, seed=0)
sub = lambda idx: HypothesisTable(table.i1[idx], table.i2[idx], table.R[idx], table.t[idx], table.theta_deg[idx], None)

def timed(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

base = RenderVerifyPipeline(model, dev, chunk=N, overlap=False, streams=1)
base.load_panos(rgb, depth)
prep = base.prepare(table)
ref = base.score(prep).clone()
torch.cuda.synchronize()
print(f"{scene}: product schedule (one stream, one chunk of {N}): {timed(lambda: base.score(prep)):.2f} ms per step", flush=True)
inw = prep["in_window"].cpu().numpy()[prep["rank"]]
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for F, pick in ((512, "last"), (512, "cheapest"), (1024, "cheapest"), (256, "cheapest")):
    idx = np.arange(N)
    fill = idx[N - F:] if pick == "last" else np.sort(np.argsort(inw, kind="stable")[:F])
    main_idx = np.setdiff1d(idx, fill)
    pa = RenderVerifyPipeline(model, dev, chunk=N - F, overlap=False, streams=1)
    pb = RenderVerifyPipeline(model, dev, chunk=F, overlap=False, streams=1)
    for p in (pa, pb):
        p.load_panos(rgb, depth)
    joint = torch.zeros((N, pa.ras.crop, pa.ras.crop, pa.engine.in_channels), dtype=torch.float16, device=dev)
    pa.tile_bufs[0], pb.tile_bufs[0] = joint[:N - F], joint[N - F:]
    prepa, prepb = pa.prepare(sub(main_idx)), pb.prepare(sub(fill))
    torch.cuda.synchronize()
    sh, sl = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0)
    out = torch.empty((N, 2), dtype=torch.float32, device=dev)

    def step():
        cur = torch.cuda.current_stream(dev)
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        for p, pr, n, st, ev in ((pa, prepa, N - F, sh, ea), (pb, prepb, F, sl, eb)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                p._scatter_chunk(pr, 0, n, 0, 0)
                p._densify_chunk(pr, 0, n, 0, 0)
                ev.record(st)
        cur.wait_event(ea); cur.wait_event(eb)
        pa.engine.forward_nhwc(joint, out=out)

    step()
    torch.cuda.synchronize()
    got = torch.empty_like(ref)
    got[torch.from_numpy(main_idx).to(dev)] = out[:N - F]
    got[torch.from_numpy(fill).to(dev)] = out[N - F:]
    assert torch.equal(got, ref), "logits differ"
    print(f"{scene}: main {N - F} (high priority) + filler {F} ({pick}, low priority): {timed(step):.2f} ms per step", flush=True)
    del pa, pb, joint
    torch.cuda.empty_cache()
