"""General-walk work counters (GPU box only).  Builds the library with -DSALVE_PROFILE_WALK into tools/ (run once HERE
with --build-only so that the .so travels), loads it in place of the product library and prints the per-render means."""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
SO = ROOT / "tools" / "libsalve_profile.so"
if "--build-only" in sys.argv:
    srcs = sorted(str(p) for p in (ROOT / "salve_amd" / "csrc").glob("*.hip"))
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DSALVE_PROFILE_WALK"] + (["-DSALVE_PROFILE_NO_COUNT"] if "--no-count" in sys.argv else []) + [
                    "-fPIC", "-shared", "-o", str(SO)] + srcs, check=True)
    sys.exit(0)
import numpy as np, torch
from salve_amd import _lib
_lib.LIB_PATH = SO
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
n = 64
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i) for i in range(4)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, 4, seed=0)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[:n], np.zeros(n), hyp.R[:n], hyp.t[:n], np.ones(n)))
bev, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
torch.cuda.synchronize()
st = dbg.stats.cpu().numpy().astype(np.float64)
names = ["apex", "apex_slow", "apex_far", "rows", "bits", "exact", "apex_table", "apex_cached"]
print({k: round(v / (1 if k in ("rows", "bits") else 64), 1) for k, v in zip(names, st.mean(0))}, "(per render; queries are counted by all 64 lanes: / 64; rows and bits are lane counts)")
ras.cfg.reserved1 = 64
bev, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
torch.cuda.synchronize()
st = dbg.stats.cpu().numpy().astype(np.float64) * 16
names = ["table", "window", "share", "slow", "far", "e2_total", "emit (E2 raster)", "nearest"]
print("wave-cycles per render (sum over 8 waves):", {k: round(v) for k, v in zip(names, st.mean(0))})
ras.cfg.reserved1 = 128
bev, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
torch.cuda.synchronize()
st = dbg.stats.cpu().numpy().astype(np.float64) * 16
st[:, 6:] /= 16
names = ["A-D (sites, mask)", "E1 wait", "E2", "F", "G", "E1 (own work)", "hard: long edge", "hard: table exhausted"]
print("phases, wave-cycles per render (sum over 8 waves):", {k: round(v) for k, v in zip(names, st.mean(0))})
