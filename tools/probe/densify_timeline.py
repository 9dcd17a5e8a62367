"""Development (GPU box): the timeline of ONE densify launch of 4096 renders -- when every render's workgroup started and ended (wall clock,
tools/probe/ablations/densify_timeline.patch built into tools/probe/_abl/libsalve_timeline.so: two stamps per workgroup in the unused end of its hard-list
region) -- for the costly-first order and the given order.  Prints the launch's span, the workgroups in flight over time, how long the tail
is, how well the cost count predicts a render's duration, and what a perfect longest-first order would have given (list-scheduling simulation
on 512 slots with the measured durations).   usage: SALVE_HIP_LIB=tools/probe/_abl/libsalve_timeline.so python tools/probe/densify_timeline.py [scene]"""
import os, sys, heapq
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
assert "timeline" in os.environ.get("SALVE_HIP_LIB", ""), "needs the timeline build (see the docstring)"
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
scene = sys.argv[1] if len(sys.argv) > 1 else "box"
dev = torch.device("cuda:0")
N, P = 4096, 64
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i, scene=scene) for i in range(P)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(N, P, seed=0)
o = np.argsort(hyp.i1, kind="stable")
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[o], np.zeros(N), hyp.R[o], hyp.t[o], np.ones(N)))
buf = torch.empty((N, *ras.bev_hw), dtype=torch.int32, device=dev)
H, W = ras.bev_hw
npx = H * W

def stamps():
    ws = ras._workspace(N)
    base = (-ws.data_ptr()) % 256
    hard0 = base + N * npx * 12                      # triq 8 B / pixel, site list 4 B / pixel, then the hard lists
    v = ws[hard0: hard0 + N * npx * 4].view(torch.int64).view(N, npx // 2) if (hard0 % 8 == 0 and npx % 2 == 0) else None
    if v is not None:
        t = v[:, npx // 2 - 3: npx // 2].cpu().numpy()
    else:   # odd pixel count: the stamps of render r lie at byte hard0 + r * npx * 4 + (npx - 6) * 4
        t = np.stack([ws[hard0 + r * npx * 4 + (npx - 6) * 4: hard0 + r * npx * 4 + npx * 4].cpu().numpy().view(np.int64) for r in range(N)])
    return t[:, 0].astype(np.float64), t[:, 1].astype(np.float64), t[:, 2].astype(np.int64)

def simulate(dur, order, slots=512):
    heap = [0.0] * slots
    heapq.heapify(heap)
    end = 0.0
    for r in order:
        t0 = heapq.heappop(heap)
        heapq.heappush(heap, t0 + dur[r])
        end = max(end, t0 + dur[r])
    return end

for flag, name in ((4, "as given (by panorama)"), (0, "costly first (product)")):
    ras.cfg.out_flags = flag
    for rep in range(3):
        ras.scatter(d_rgb, d_depth, hd, N, buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ras.densify(N, buf); e1.record()
        torch.cuda.synchronize()
    t0, t1, pos = stamps()
    tick = 1e-5   # wall_clock64: 100 MHz -> ms
    s, e = (t0 - t0.min()) * tick, (t1 - t0.min()) * tick
    dur = e - s
    span = e.max()
    print(f"== {scene}, {name}: HIP events {e0.elapsed_time(e1):.2f} ms; first start -> last end {span:.2f} ms; a render's workgroup lives {np.median(dur):.2f} ms (median), "
          f"{dur.min():.2f} .. {dur.max():.2f}; sum of lifetimes / 512 slots = {dur.sum() / 512:.2f} ms")
    grid = np.linspace(0, span, 29)
    infl = [(int(((s <= g) & (e > g)).sum())) for g in grid]
    print("   workgroups in flight at 28 equal steps:", " ".join(str(v) for v in infl))
    last_full = max(g for g, v in zip(np.linspace(0, span, 2001), [int(((s <= g) & (e > g)).sum()) for g in np.linspace(0, span, 2001)]) if v >= 500)
    print(f"   fewer than 500 in flight from {last_full:.2f} ms on: a tail of {span - last_full:.2f} ms; idle slot-time in it {sum(max(0.0, span - max(x, last_full)) for x in e) / 512:.2f} ms-equivalents")
    by_pos = np.argsort(pos)
    print(f"   list scheduling of the measured lifetimes on 512 slots: this order {simulate(dur, by_pos):.2f} ms, longest first {simulate(dur, np.argsort(-dur)):.2f} ms, "
          f"shortest first {simulate(dur, np.argsort(dur)):.2f} ms")
ras.cfg.out_flags = 0
# ---- how well do counts from the occupancy bitmap predict a render's lifetime?  (features from the sparse images, torch on the device)
ras.scatter(d_rgb, d_depth, hd, N, buf)
torch.cuda.synchronize()
F = {k: np.zeros(N) for k in ("o4: a 4 x 9 half-window empty", "o6: a 6 x 13 half-window empty", "o4 both: two opposite half-windows empty", "sites", "b1: a 4-neighbour missing", "b2: >= 2 of 4 missing", "b3: >= 5 of 8 missing", "b4: >= 3 of 4 missing", "b5: all 8 missing")}
for lo in range(0, N, 256):
    occ = (buf[lo:lo + 256] != 0)
    pad = torch.nn.functional.pad(occ, (1, 1, 1, 1))
    nb = pad[:, :-2, 1:-1].int() + pad[:, 2:, 1:-1].int() + pad[:, 1:-1, :-2].int() + pad[:, 1:-1, 2:].int()
    nb8 = nb + pad[:, :-2, :-2].int() + pad[:, :-2, 2:].int() + pad[:, 2:, :-2].int() + pad[:, 2:, 2:].int()
    cnt = lambda m: m.flatten(1).sum(1).cpu().numpy()
    of = occ.float()[:, None]
    def half_empty(r, h):   # for every pixel: is the h x (2r+1) window to its right / left / below / above free of sites?
        kx = torch.ones((1, 1, 2 * r + 1, h), device=dev)
        ky = torch.ones((1, 1, h, 2 * r + 1), device=dev)
        px = torch.nn.functional.pad(of, (h, h, r, r))
        py = torch.nn.functional.pad(of, (r, r, h, h))
        cx = torch.nn.functional.conv2d(px, kx)   # [.., H, W + h + 1]: window starting at column x - h + k
        cy = torch.nn.functional.conv2d(py, ky)
        Wd, Hd = of.shape[-1], of.shape[-2]
        right = cx[..., :, h + 1: h + 1 + Wd] == 0
        left = cx[..., :, 0: Wd] == 0
        below = cy[..., h + 1: h + 1 + Hd, :] == 0
        above = cy[..., 0: Hd, :] == 0
        return right[:, 0], left[:, 0], below[:, 0], above[:, 0]
    r4 = half_empty(4, 4)
    r6 = half_empty(6, 6)
    F["o4: a 4 x 9 half-window empty"][lo:lo + 256] = cnt(occ & (r4[0] | r4[1] | r4[2] | r4[3]))
    F["o6: a 6 x 13 half-window empty"][lo:lo + 256] = cnt(occ & (r6[0] | r6[1] | r6[2] | r6[3]))
    F["o4 both: two opposite half-windows empty"][lo:lo + 256] = cnt(occ & ((r4[0] & r4[1]) | (r4[2] & r4[3])))
    F["sites"][lo:lo + 256] = cnt(occ)
    F["b1: a 4-neighbour missing"][lo:lo + 256] = cnt(occ & (nb < 4))
    F["b2: >= 2 of 4 missing"][lo:lo + 256] = cnt(occ & (nb <= 2))
    F["b3: >= 5 of 8 missing"][lo:lo + 256] = cnt(occ & (nb8 <= 3))
    F["b4: >= 3 of 4 missing"][lo:lo + 256] = cnt(occ & (nb <= 1))
    F["b5: all 8 missing"][lo:lo + 256] = cnt(occ & (nb8 == 0))
ras.densify(N, buf)
torch.cuda.synchronize()
t0, t1, pos = stamps()
dur = (t1 - t0) * 1e-5
print(f"== {scene}: predicting a render's lifetime (ms) from bitmap counts; list-scheduling makespan on 512 slots with the renders ordered by the prediction "
      f"(perfect knowledge: {simulate(dur, np.argsort(-dur)):.2f} ms, unordered: {simulate(dur, np.arange(N)):.2f} ms)")
for k, v in F.items():
    print(f"   {k:28s} correlation {np.corrcoef(v, dur)[0, 1]:.3f}   ordered by it: {simulate(dur, np.argsort(-v)):.2f} ms")
X = np.stack([np.ones(N)] + list(F.values()), 1)
w, *_ = np.linalg.lstsq(X, dur, rcond=None)
pred = X @ w
print("   least squares over all six + constant: weights", " ".join(f"{x:.3g}" for x in w), f"correlation {np.corrcoef(pred, dur)[0, 1]:.3f}   ordered by it: {simulate(dur, np.argsort(-pred)):.2f} ms")
for ks in (("b1: a 4-neighbour missing", "o4: a 4 x 9 half-window empty"), ("b1: a 4-neighbour missing", "o6: a 6 x 13 half-window empty"), ("b2: >= 2 of 4 missing", "b5: all 8 missing"), ("b1: a 4-neighbour missing", "b3: >= 5 of 8 missing"), ("sites", "b2: >= 2 of 4 missing", "b4: >= 3 of 4 missing")):
    X2 = np.stack([np.ones(N)] + [F[k] for k in ks], 1)
    w2, *_ = np.linalg.lstsq(X2, dur, rcond=None)
    p2 = X2 @ w2
    print(f"   least squares over {ks}: weights", " ".join(f"{x:.3g}" for x in w2), f"correlation {np.corrcoef(p2, dur)[0, 1]:.3f}   ordered by it: {simulate(dur, np.argsort(-p2)):.2f} ms")
