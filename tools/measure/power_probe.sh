#!/bin/bash
# Development: board power and shader clock (rocm-smi) sampled while the ResNet-50 forward at batch 4096 runs in a loop, for the product
# library and for each library given on the command line.   usage: power_probe.sh [<lib> ...]
cd $GRAFT_REPO_ROOT
run() {  # $1 = label, SALVE_HIP_LIB set by the caller
  python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from types import SimpleNamespace
from salve_amd.models.early_fusion import EarlyFusionCEResnet
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
eng = m.compiled(dev)
x = torch.randn(4096, 224, 224, eng.in_channels, device=dev).to(torch.float16)
eng.forward_nhwc(x); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 8.0:
    for _ in range(10): eng.forward_nhwc(x)
    torch.cuda.synchronize(); n += 10
print(f"   forward {1e3 * (time.perf_counter() - t0) / n:.2f} ms", flush=True)
PY
  local pid=$!
  sleep 5
  for k in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk" | tr -s ' ' | sed "s/^/   [$1] /"; sleep 0.7; done
  wait $pid
}
echo "== product"; run product
for V in "$@"; do echo "== $V"; SALVE_HIP_LIB=$V run $(basename $V); done
