#!/bin/bash
# Development: board power and shader clock (rocm-smi) while ONE kernel shape runs in a loop (tools/measure/bench_conv.py at batch 4096), shape by
# shape, then the rasteriser's scatter + densify.  Is the chip at its power cap, and in which kernels?
cd $GRAFT_REPO_ROOT
sample() { for k in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | tr -s ' \t' ' ' | sed 's/GPU\[0\] : //' | tr '\n' ' '; echo; sleep 0.6; done; }
while IFS='|' read -r name reps; do
  echo "== $name"
  SALVE_BENCH_ONLY="$name" SALVE_BENCH_REPS=$reps timeout -k 10 120 python tools/measure/bench_conv.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-110 &
  pid=$!; sleep 6; sample; wait $pid
done <<'LIST'
l3.conv2 3x3 256>256 @14|9000
l2.conv2 3x3 128>128 @28|8000
l4.conv1 1x1 2048>512 @7|20000
l1.conv1 1x1 256>64 @56|5000
l2.conv3 1x1 128>512 +res|4000
LIST
echo "== rasteriser: scatter + densify, 4096 renders"
python - <<'PY' &
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
dev = torch.device("cuda:0"); ras = BevRasteriser(dev); n, P = 4096, 64
panos = [synthetic.make_pano(i) for i in range(P)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, P, seed=0); o = np.argsort(hyp.i1, kind="stable")
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[o], np.zeros(n), hyp.R[o], hyp.t[o], np.ones(n)))
bev = torch.empty((n,) + ras.bev_hw, dtype=torch.int32, device=dev)
ras.scatter(d_rgb, d_depth, hd, n, bev); ras.densify(n, bev); torch.cuda.synchronize()
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 9.0:
    for _ in range(5): ras.scatter(d_rgb, d_depth, hd, n, bev); ras.densify(n, bev)
    torch.cuda.synchronize(); k += 5
print(f"   scatter + densify {1e3 * (time.perf_counter() - t0) / k:.2f} ms per 4096", flush=True)
PY
pid=$!; sleep 12; sample; wait $pid
