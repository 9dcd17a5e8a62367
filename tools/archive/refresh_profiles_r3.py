"""Turn gpurun_out/r3s (tools/gpu_r3s.sh) and gpurun_out/r3traffic (tools/gpu_r3_traffic.sh) into the tracked round-3 summaries under
profiles/ and refresh profiles/traffic.json (which bench.py reads for its `traffic` fields)."""
import collections, csv, glob, json, shutil, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "gpurun_out" / "r3s"
TRF = ROOT / "gpurun_out" / "r3traffic"
PRO = ROOT / "profiles"
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
line = lambda f: json.loads([x for x in open(SRC / f) if x.startswith("{")][-1])
VER = ("conv_igemm", "conv8_kernel", "bottleneck", "stem_pool", "maxpool", "avgpool", "expand_chain")

import os
def newest(pattern, recursive=False):
    # gpurun merges a call's files INTO gpurun_out/ without deleting those of earlier calls, and rocprofv3 names its files after
    # the process id: take the newest match, never all of them
    files = glob.glob(pattern, recursive=recursive)
    return max(files, key=os.path.getmtime)

def stats(d, title, tag, n_rows=16):
    f = newest(str(SRC / d / "*" / "*kernel_stats.csv"))
    shutil.copy(f, PRO / f"r03_bench_kernel_stats{tag}.csv")
    rows = list(csv.DictReader(open(f)))
    out = title + ["", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:n_rows]:
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return out, rows

lines = {k: line(f) for k, f in (("default", "bench.log"), ("two_streams", "bench_2s.log"), ("three_streams", "bench_3s.log"),
                                  ("cluttered_scene", "bench_cluttered.log"), ("config5", "bench_c5.log"), ("rccl_world1", "bench_rccl.log"))}
for k, v in lines.items():
    (PRO / ("r03_bench_line.json" if k == "default" else f"r03_bench_line_{k}.json")).write_text(json.dumps(v) + "\n")
b1 = lines["default"]
n_launch = 3  # warm-up + 2 steps
o1, rows1 = stats("prof1", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline   (round 3, MI355X: the benchmark's defaults,",
                            "# 4096 hypotheses in ONE launch per stage, one HIP stream)",
                            f"# bench line of the same build without the profiler: r03_bench_line.json ({b1['value'] / 1e3:.1f} k hypotheses/s)"], "")
d = [r for r in rows1 if "bev_densify" in r["Name"]][0]
sc = [r for r in rows1 if "bev_scatter_kernel" in r["Name"]][0]
ver = sum(int(r["TotalDurationNs"]) for r in rows1 if any(k in r["Name"] for k in VER)) / 1e6 / n_launch
o1 += ["", f"bev_densify_kernel: {d['Calls']} launches = {n_launch} of 4096 renders + 1 of 64 (identity renders): {(int(d['TotalDurationNs']) / 1e6 - 0.8) / n_launch:.2f} ms per 4096 renders;",
       f"bev_scatter_kernel: {sc['Calls']} launches (two passes each): {int(sc['TotalDurationNs']) / 1e6 / (n_launch + 64 / 4096):.2f} ms per 4096 renders;",
       f"verifier kernels: {ver:.2f} ms per 4096 samples = {4096 * 8.41 / ver:.0f} TFLOP/s = {4096 * 8.41 / ver / 25:.1f} % of the 2.5 PFLOP/s dense fp16 peak.",
       f"bench.py's live HIP events of its own (un-profiled) run: scatter {b1['roofline']['scatter_ms']} ms, densify {b1['roofline']['densify_ms']} ms, verifier {b1['roofline_verifier']['launch_ms']} ms."]
(PRO / "r03_bench_kernel_stats.md").write_text("\n".join(o1) + "\n")
o3, rows3 = stats("prof3", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 3   (same build: scatter | densify | verifier",
                            "# on three streams, the rasteriser of pass k + 1 under the verifier of pass k; per-kernel durations include the slow-down from sharing the CUs)",
                            f"# bench line: r03_bench_line_three_streams.json ({lines['three_streams']['value'] / 1e3:.1f} k hypotheses/s)"], "_three_streams")
(PRO / "r03_bench_kernel_stats_three_streams.md").write_text("\n".join(o3) + "\n")
o5, rows5 = stats("prof5", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --pano-hw 1024x2048 --surfaces floor,ceiling --layers 152 --hyps 1024 --panos 16 --chunk 512 --steps 2 --warmup 1",
                            "# BASELINE config 5 on ONE GPU (2048x1024 panoramas, floor + ceiling, ResNet-152 with 12 input channels, fp16)",
                            f"# bench line: r03_bench_line_config5.json ({lines['config5']['value'] / 1e3:.1f} k hypotheses/s)"], "_config5", 20)
(PRO / "r03_bench_kernel_stats_config5.md").write_text("\n".join(o5) + "\n")

# ---- overlap table: one / two / three streams, per-kernel durations serial vs three streams, footprints
def avg(rows, name):
    r = [x for x in rows if name in x["Name"]]
    return sum(int(x["TotalDurationNs"]) for x in r) / max(1, sum(int(x["Calls"]) for x in r)) / 1e3
kn = ["bev_scatter_kernel", "bev_densify_kernel", "bev_tile_pair_kernel", "stem_pool_kernel", "bottleneck_kernel<64, 8, false>", "conv_igemm_kernel<128, true, false>",
      "conv_igemm_kernel<128, false, false>", "conv8_kernel<false, false>", "conv8_kernel<true, false>", "expand_chain_kernel<128, 128", "expand_chain_kernel<256, 256, 5"]
ov = ["# Stream overlap, round 3 (MI355X, the benchmark's workload: 4096 hypotheses per launch; bench lines profiles/r03_bench_line*.json)", "",
      "| schedule | hypotheses/s | ms per step |", "|---|---|---|"]
for k, lab in (("default", "one stream (the default)"), ("two_streams", "two streams: rasteriser | verifier"), ("three_streams", "three streams: scatter | densify | verifier")):
    ov.append(f"| {lab} | {lines[k]['value']:.0f} | {lines[k]['ms_per_step']:.2f} |")
ov += ["", "Average launch durations (rocprofv3 --kernel-trace --stats, same commands under the profiler):", "",
       "| kernel | one stream, us | three streams, us | ratio |", "|---|---|---|---|"]
for name in kn:
    a, b = avg(rows1, name), avg(rows3, name)
    if a > 0:
        ov.append(f"| {name} | {a:.0f} | {b:.0f} | {b / a:.2f} |")
(PRO / "r03_overlap.md").write_text("\n".join(ov) + "\n")

# ---- rasteriser traffic at the benchmark's launch shape (4096 renders per launch, 64 panoramas)
def pmc(sub, counter):
    acc = collections.defaultdict(list)
    for f in [newest(str(SRC / sub / "**" / "*counter_collection.csv"), recursive=True)]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "bev_" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
n = 4096
rows = []
tot = 0.0
for k in ("bev_scatter_kernel", "bev_densify_kernel"):
    per_pass = 2 if "scatter" in k else 1     # scatter: two passes per render launch
    f = sum(fe[k]) / len(fe[k]) * per_pass
    w = sum(wr[k]) / len(wr[k]) * per_pass
    rows.append((k, f, w))
    tot += f + w
per_render = tot * 1024 / n
txt = ["# HBM traffic counters of the rasteriser, round 3 (MI355X, rocprofv3 --pmc, one counter per pass, with --kernel-trace only)", "",
       "Command (per pass): `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/pmc_render.py 4096 64`",
       "-- the BENCHMARK's launch shape: 4096 renders per launch over 64 panoramas of 1024x512 -> 501x501 BEV (round 2 measured 512 renders over 4 panoramas).",
       "Units: the counters are in KiB.  FETCH_SIZE is reported as counted: the guide's x2 correction holds for wide (16 B / lane) coalesced streams; the",
       "rasteriser's loads are 2 - 8 bytes per lane, for which the counter is uncalibrated -- the read side is a lower bound, as in rounds 1 and 2.", "",
       "| kernel (per launch of 4096 renders) | FETCH_SIZE KiB | WRITE_SIZE KiB |", "|---|---|---|"]
for k, f, w in rows:
    txt.append(f"| `{k}`{' (pass 0 + pass 1)' if 'scatter' in k else ''} | {f:,.0f} | {w:,.0f} |")
txt += [f"| **whole rasteriser** | **{sum(r[1] for r in rows):,.0f}** | **{sum(r[2] for r in rows):,.0f}** |", "",
        f"Per render: {per_render / 1e6:.2f} MB against 2.555 MB algorithmic (SURVEY 8d): {per_render / 2555243:.2f}x (5.56 MB in round 2 at 512 renders per launch over 4 panoramas:",
        "with 64 panoramas fewer panorama lines are shared in L2).  bench.py reads this figure from profiles/traffic.json."]
(PRO / "r03_pmc_traffic.md").write_text("\n".join(txt) + "\n")
shutil.copy(SRC / "ras_traffic.txt", PRO / "r03_pmc_traffic_raw.txt")
shutil.copy(SRC / "ras_sq.txt", PRO / "r03_densify_sq_raw.txt")

# ---- verifier traffic table (tools/gpu_r3_traffic.sh -> tools/resnet_traffic_report.py)
shutil.copy(TRF / "report.md", PRO / "r03_resnet_traffic.md")
s = json.load(open(TRF / "summary.json"))
B = s["batch"]
t = {"_comment": "Counted HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes) per unit of work, at the launch shape named by the key; written by tools/refresh_profiles_r3.py from gpurun_out/. bench.py reads this file; a workload without an entry reports traffic: null.",
     f"rasteriser/1024x512/launch{n}": {"bytes_per_unit": per_render, "unit": "render", "source": "profiles/r03_pmc_traffic.md"},
     f"verifier/resnet50-6ch/launch{B}": {"bytes_per_unit": (s["counted_read_bytes"] + s["counted_write_bytes"]) / B, "unit": "sample", "source": "profiles/r03_resnet_traffic.md"},
     "verifier_algorithmic/resnet50-6ch": {"bytes_per_unit": s["alg_bytes"] / B, "unit": "sample", "source": "profiles/r03_resnet_traffic.md (activations in + residual + out per launch at the present fusion level, fp16, + weights)"}}
json.dump(t, open(PRO / "traffic.json", "w"), indent=1)
print("\n".join(o1[:24])); print("\n".join(ov)); print("\n".join(txt[-4:]))
