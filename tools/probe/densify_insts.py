"""Which phase of bev_densify_kernel issues the instructions: the kernel is launched whole and with phases switched off through
the development flags (as tools/probe/densify_ablation.py does for times), one launch each after a warm-up, under
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU`.
With a directory argument it prints the counters per launch instead (report mode).
  run:    rocprofv3 ... -d OUT -- python3 tools/probe/densify_insts.py 2048
  report: python3 tools/probe/densify_insts.py --report OUT 2048"""
import sys
from pathlib import Path
FLAGS = (("whole kernel", 0), ("E2 walks, no rasterisation inside", 2), ("no general walk (E2)", 4), ("no triangle rasterisation (F)", 8),
         ("no E2, no F", 12), ("no E2, no F, lean walks queue nothing", 12 | 1024), ("no star walk at all (B, B2, C, G only)", 1))
if sys.argv[1] == "--report":
    import csv, glob, collections
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    rows = collections.defaultdict(dict)
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bev_densify_kernel" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                rows[int(r["Dispatch_Id"])]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(rows)[-len(FLAGS):]
    for (name, _), i in zip(FLAGS, ids):
        v = rows[i]
        print(f"{name:42s} {v['us'] / n:6.2f} us/render  " + "  ".join(f"{k.replace('SQ_', '')}={v[k] / n / 1e3:8.1f}k" for k in sorted(v) if k != "us"))
    sys.exit(0)
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
ras.render(d_rgb, d_depth, hd, n); torch.cuda.synchronize()
for _, fl in FLAGS:
    ras.cfg.reserved1 = fl
    ras.render(d_rgb, d_depth, hd, n); torch.cuda.synchronize()
