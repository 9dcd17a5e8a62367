"""Development (CPU only): the site sets of a few synthetic renders (oracle, exact mode) as int32 (x, y) pairs, for tools/probe/host/walk_counters.cpp.
usage: python tools/probe/host/walk_sites.py [out_dir = /tmp/walk_sites]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
import numpy as np
from oracle import bev_oracle as bo
from salve_amd import synthetic
out = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/walk_sites")
out.mkdir(parents=True, exist_ok=True)
hyp = synthetic.make_hypotheses(16, 2, seed=0)
for scene in ("box", "cluttered", "noisy"):
    for j in (0, 5):
        rgb, depth = synthetic.make_pano(j % 2, scene=scene)
        a = bo.xyzrgb_from_arrays(depth, rgb, bo.floor_ceiling_z_range("floor"))
        a, _ = bo.pose_pair(a, a[:1], hyp.R[j], hyp.t[j])
        sp = bo.render_bev_image(a, mode="exact")["site_xy_sorted"].astype(np.int32)
        sp.tofile(out / f"sites_{scene}_{j}.bin")
        print(scene, j, len(sp), "sites")
