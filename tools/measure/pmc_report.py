"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs.  usage: pmc_report.py <dir> [name filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt and flt not in n: continue
        key = (n, r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[key] = (r["VGPR_Count"], r["LDS_Block_Size"], r["SGPR_Count"])
for key in sorted(acc):
    v = acc[key]
    print(f"{key[0][:60]} grid={key[1]} vgpr={meta[key][0]} lds={meta[key][1]} n={len(next(iter(v.values())))}")
    print("   " + "  ".join(f"{c.replace('SQ_', '')}={sum(x) / len(x):.4g}" for c, x in sorted(v.items())))
