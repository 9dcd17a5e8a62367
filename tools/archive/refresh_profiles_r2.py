"""Turn gpurun_out/r2s (tools/gpu_r2s.sh) into the tracked round-2 summaries under profiles/."""
import collections, csv, glob, json, shutil, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "gpurun_out" / "r2s"
PRO = ROOT / "profiles"
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
line = lambda f: json.loads([x for x in open(SRC / f) if x.startswith("{")][0])

def stats(d, title, tag):
    f = glob.glob(str(SRC / d / "*" / "*kernel_stats.csv"))[0]
    shutil.copy(f, PRO / f"r02_bench_kernel_stats{tag}.csv")
    rows = list(csv.DictReader(open(f)))
    out = title + ["", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:14]:
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return out, rows

b1, b3, bc = line("bench.log"), line("bench_3s.log"), line("bench_cluttered.log")   # default (one stream), --streams 3, cluttered scene
(PRO / "r02_bench_line.json").write_text(json.dumps(b1) + "\n")
(PRO / "r02_bench_line_three_streams.json").write_text(json.dumps(b3) + "\n")
(PRO / "r02_bench_line_cluttered_scene.json").write_text(json.dumps(bc) + "\n")
o1, rows1 = stats("prof1", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline   (round 2, MI355X: the benchmark's defaults,",
                            "# 4096 hypotheses in ONE launch per stage, one HIP stream)",
                            f"# bench line of the same build without the profiler: r02_bench_line.json ({b1['value'] / 1e3:.1f} k hypotheses/s)"], "")
n_launch = 3  # warm-up + 2 steps
d = [r for r in rows1 if "bev_densify" in r["Name"]][0]
sc = [r for r in rows1 if "bev_scatter_kernel" in r["Name"]][0]
ver = sum(int(r["TotalDurationNs"]) for r in rows1 if any(k in r["Name"] for k in ("conv_igemm", "conv8_kernel", "conv_wide", "conv_pc", "bottleneck", "stem_pool", "maxpool", "avgpool"))) / 1e6 / n_launch
o1 += ["", f"bev_densify_kernel: {d['Calls']} launches = {n_launch} of 4096 renders + 1 of 64 (identity renders): {(int(d['TotalDurationNs']) / 1e6 - 0.8) / n_launch:.2f} ms per 4096 renders;",
       f"bev_scatter_kernel: {sc['Calls']} launches (two passes each): {int(sc['TotalDurationNs']) / 1e6 / (n_launch + 64 / 4096):.2f} ms per 4096 renders;",
       f"verifier kernels: {ver:.2f} ms per 4096 samples = {4096 * 8.41 / ver:.0f} TFLOP/s = {4096 * 8.41 / ver / 25:.1f} % of the 2.5 PFLOP/s dense fp16 peak.",
       f"bench.py's live HIP events of its own (un-profiled) run: scatter {b1['roofline']['scatter_ms']} ms, densify {b1['roofline']['densify_ms']} ms, verifier {b1['roofline_verifier']['launch_ms']} ms."]
(PRO / "r02_bench_kernel_stats.md").write_text("\n".join(o1) + "\n")
o, rows = stats("prof3", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 3   (same build: scatter | densify | verifier",
                          "# on three streams, the rasteriser of pass k + 1 under the verifier of pass k; per-kernel durations include the slow-down from sharing the CUs)",
                          f"# bench line: r02_bench_line_three_streams.json ({b3['value'] / 1e3:.1f} k hypotheses/s)"], "_three_streams")
(PRO / "r02_bench_kernel_stats_three_streams.md").write_text("\n".join(o) + "\n")

# ---- verifier SQ counters + per-launch trace (tools/trace_resnet.py, batch 512, last of three forwards)
def counters(d):
    f = glob.glob(str(SRC / d / "**" / "*counter_collection.csv"), recursive=True)[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    ks = [v for v in disp.values() if any(k in v["name"] for k in ("conv_igemm", "conv8_kernel", "conv_wide", "conv_pc", "bottleneck", "stem_pool", "maxpool", "avgpool"))]
    return ks[-(len(ks) // 3):]
a, b, c = counters("sq1"), counters("sq2"), counters("sq3")
assert len(a) == len(b) == len(c)
import subprocess
tr = subprocess.run([sys.executable, str(ROOT / "tools" / "trace_resnet_report.py"), str(SRC / "trace")], capture_output=True, text=True).stdout.splitlines()
out = ["# SQ counters of the verifier, round 2 (MI355X, rocprofv3 --pmc, --kernel-trace only; tools/trace_resnet.py: ResNet-50, 6 channels, batch 512)",
       "", "Passes: (1) SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES,",
       "(2) SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM, (3) GRBM_GUI_ACTIVE.",
       "MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of SIMD cycles in which the matrix pipe executes",
       "(v_mfma_f32_16x16x32_f16 = 16 cycles each; the guide's units).  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier),",
       "stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls), LDS = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles), conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.",
       "", "| # | kernel | us (profiled) | MFMA-busy | wait | stall | LDS busy | LDS conflicts | clock GHz |", "|---|---|---|---|---|---|---|---|---|"]
tot_mfma = tot_cyc = tot_us = 0.0
for i, (x, y, z) in enumerate(zip(a, b, c)):
    us = (x["end"] - x["start"]) / 1e3
    cyc = z["GRBM_GUI_ACTIVE"] / 8.0
    us3 = (z["end"] - z["start"]) / 1e3
    mf = x["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) if cyc else 0
    tot_mfma += x["SQ_VALU_MFMA_BUSY_CYCLES"]; tot_cyc += cyc; tot_us += us
    lds = y["SQ_LDS_IDX_ACTIVE"] / (256 * cyc) if cyc else 0
    conf = y["SQ_LDS_BANK_CONFLICT"] / y["SQ_LDS_IDX_ACTIVE"] if y["SQ_LDS_IDX_ACTIVE"] else 0
    out.append(f"| {i} | {x['name'][:44]} | {us:.0f} | {100 * mf:.0f} % | {100 * x['SQ_WAIT_ANY'] / x['SQ_WAVE_CYCLES']:.0f} % | {100 * x['SQ_WAIT_INST_ANY'] / x['SQ_WAVE_CYCLES']:.0f} % | {100 * lds:.0f} % | {100 * conf:.0f} % | {cyc / us3 / 1e3:.2f} |")
out += ["", f"Whole forward: {tot_us / 1e3:.2f} ms under the profiler, MFMA-busy {100 * tot_mfma / (1024 * tot_cyc):.1f} % of the SIMD cycles (time-weighted);",
        "algorithmic: 512 x 8.41 GFLOP over the un-profiled forward time (tools/bench_resnet.py) -> the TFLOP/s quoted in DESIGN.md.", "",
        "Per-launch durations of the same forward without counters (rocprofv3 --kernel-trace only):", "```"] + [l.rstrip() for l in tr] + ["```"]
(PRO / "r02_resnet_sq.md").write_text("\n".join(out) + "\n")
print("\n".join(out[:70]))
