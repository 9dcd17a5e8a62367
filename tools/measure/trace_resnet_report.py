"""Per-launch table from a rocprofv3 kernel trace of tools/measure/trace_resnet.py: the LAST forward's launches in launch order
(name, grid, duration).  The op program is printed next to it by tools/measure/trace_resnet.py --ops."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv8_kernel", "conv_wide", "conv_pc", "stem_pool", "maxpool", "avgpool", "bottleneck", "expand_chain"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_per = len(rows) // 3   # tools/measure/trace_resnet.py runs three forwards; the last one is reported
tot = 0
for i, r in enumerate(rows[-n_per:]):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{i:3d} {name:42s} wgs={int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)))):7d} {dur:9.1f} us")
print("total", round(tot, 1), "us")
