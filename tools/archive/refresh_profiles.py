"""Turn the rocprofv3 --stats outputs of a bench run into the tracked summaries under profiles/.
usage: python tools/refresh_profiles.py <tag>   (reads gpurun_out/prof_<tag>, prof_<tag>_serial, bench_<tag>*.log)"""
import csv, glob, json, shutil, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
def table(d, title, out_csv):
    f = glob.glob(str(ROOT / "gpurun_out" / d / "*" / "*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    shutil.copy(f, out_csv)
    out = title + ["", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:13]:
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    return out, rows
line = lambda f: json.loads([x for x in open(ROOT / "gpurun_out" / f) if x.startswith('{')][0])
l3, l1 = line(f"bench_{tag}_prof.log"), line(f"bench_{tag}_serial.log")
o, rows = table(f"prof_{tag}", ["# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (round 1, MI355X, 4096 hypotheses, chunk 1024, 2 steps + 1 warm-up)",
    "# the default three-stream run: scatter, densify and verifier kernels overlap, so per-kernel durations include the slow-down from sharing the CUs",
    f"# source: gpurun_out/prof_{tag}/*/_kernel_stats.csv ; names shortened; bench line of the same run: r01_bench_line_under_rocprof.json"], ROOT / "profiles" / "r01_bench_kernel_stats.csv")
d = [r for r in rows if 'bev_densify' in r['Name']][0]
o += ["", f"bev_densify_kernel: {d['Calls']} launches = 12 of 1024 renders (4 per pass x 3 passes) + 1 of 64 (the cached identity renders);",
      f"total {int(d['TotalDurationNs'])/1e6:.2f} ms -> about {(int(d['TotalDurationNs'])/1e6-0.8)/12:.2f} ms per 1024-render launch; bench.py's live HIP-event average over the",
      f"{l3['roofline']['launches_timed']} launches of its timed region in the same run: roofline.launch_ms = {l3['roofline']['launch_ms']}."]
(ROOT / "profiles" / "r01_bench_kernel_stats.md").write_text("\n".join(o) + "\n")
o2, rows2 = table(f"prof_{tag}_serial", ["# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-overlap   (same build, ONE stream, chunk 1024)",
    f"# per-kernel durations without the other streams' kernels on the CUs; {l1['value']/1e3:.1f} k hypotheses/s in this mode ({l3['value']/1e3:.1f} k with three streams)",
    f"# source: gpurun_out/prof_{tag}_serial/*/_kernel_stats.csv"], ROOT / "profiles" / "r01_bench_kernel_stats_one_stream.csv")
(ROOT / "profiles" / "r01_bench_kernel_stats_one_stream.md").write_text("\n".join(o2) + "\n")
for src, dst in ((f"bench_{tag}.log", "r01_bench_line.json"), (f"bench_{tag}_prof.log", "r01_bench_line_under_rocprof.json")):
    (ROOT / "profiles" / dst).write_text([x for x in open(ROOT / "gpurun_out" / src) if x.startswith('{')][0])
conv = sum(int(r['TotalDurationNs']) for r in rows2 if 'conv_igemm' in r['Name'] or 'bottleneck' in r['Name']) / 1e6 / 12
print("\n".join(o2)); print("verifier convs per 1024:", round(conv, 2), "ms"); print(l3['roofline']); print(l1['roofline'])
