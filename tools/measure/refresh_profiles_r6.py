"""Turn gpurun_out/r6s (tools/measure/gpu_r6_profiles.sh) and gpurun_out/r6traffic_v50 / r6traffic_v152 (tools/measure/gpu_verifier_traffic.sh) into the
tracked round-6 summaries under profiles/ and refresh profiles/traffic.json (which bench.py reads for its `traffic` fields).

Traffic convention (ONE for both rooflines): counted bytes = FETCH_SIZE x 2 + WRITE_SIZE -- the guide's gfx950
correction (FETCH_SIZE tallies 128-byte requests at 64 bytes) -- with the raw FETCH_SIZE figure carried beside it
(`bytes_per_unit_raw_fetch`): the correction is calibrated for 16-byte-per-lane streams, the rasteriser's 2 - 8-byte loads lie
between the two."""
import collections, csv, glob, json, os, shutil, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
SRC = ROOT / "gpurun_out" / "r6s"
PRO = ROOT / "profiles"
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
line = lambda f: json.loads([x for x in open(SRC / f) if x.startswith("{")][-1])
VER = ("conv_igemm", "conv8_kernel", "bottleneck", "stem_pool", "maxpool", "avgpool", "expand_chain")

def newest(pattern, recursive=False):
    # gpurun merges a call's files INTO gpurun_out/ without deleting those of earlier calls, and rocprofv3 names its files after
    # the process id: take the newest match, never all of them
    return max(glob.glob(pattern, recursive=recursive), key=os.path.getmtime)

def stats(d, title, tag, n_rows=16):
    f = newest(str(SRC / d / "*" / "*kernel_stats.csv"))
    shutil.copy(f, PRO / f"r06_bench_kernel_stats{tag}.csv")
    rows = list(csv.DictReader(open(f)))
    out = title + ["", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:n_rows]:
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    return out, rows

def big_launches(d):
    """kernel name -> mean duration (us) of its launches with the LARGEST grid (the whole-shard launches; the 64-render identity
    launches of load_panos are left out)"""
    f = newest(str(SRC / d / "*" / "*kernel_trace.csv"))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append((int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = {}
    for k, v in acc.items():
        g = max(x[0] for x in v)
        sel = [x[1] for x in v if x[0] == g]
        out[k] = (sum(sel) / len(sel), len(sel))
    return out

lines = {k: line(f) for k, f in (("default", "bench.log"), ("cluttered_scene", "bench_cluttered.log"), ("noisy_scene", "bench_noisy.log"),
                                  ("config5", "bench_c5.log"), ("rccl_world1", "bench_rccl.log"))}
b1 = lines["default"]
n_launch = 3  # warm-up + 2 steps
o1, rows1 = stats("prof1", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-power-probe --no-cpu-baseline --no-calibration --no-config5   (round 6, MI355X: the benchmark's defaults,",
                            "# 4096 hypotheses in ONE launch per stage, one HIP stream)",
                            f"# bench line of the same build without the profiler: r06_bench_line.json ({b1['value'] / 1e3:.1f} k hypotheses/s)"], "")
big1 = big_launches("prof1")
ver = sum(int(r["TotalDurationNs"]) for r in rows1 if any(k in r["Name"] for k in VER)) / 1e6 / n_launch
o1 += ["", "Whole-shard launches only (4096 renders; the 64 identity renders of load_panos are separate launches and are left out):",
       f"bev_splat_kernel {big1['bev_splat_kernel<false>'][0] / 1e3:.2f} ms, bev_densify_kernel (its last phase writes the verifier tiles: no tile launch) {big1['bev_densify_kernel<false>'][0] / 1e3:.2f} ms per 4096;",
       f"verifier kernels: {ver:.2f} ms per 4096 samples = {4096 * 8.41 / ver:.0f} TFLOP/s = {4096 * 8.41 / ver / 25:.1f} % of the 2.5 PFLOP/s dense fp16 peak.",
       f"bench.py's live HIP events of its own (un-profiled) run: scatter stage {b1['roofline']['scatter_ms']} ms, densify {b1['roofline']['densify_ms']} ms, verifier {b1['roofline_verifier']['launch_ms']} ms.",
       "bev_pano_index_kernel (pose-independent block boxes, once per panorama set at load_panos) is outside the step."]
(PRO / "r06_bench_kernel_stats.md").write_text("\n".join(o1) + "\n")
o5, rows5 = stats("prof5", ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --pano-hw 1024x2048 --surfaces floor,ceiling --layers 152 --panos 16 --steps 2 --warmup 1 (4096 hypotheses = 8192 renders per launch)",
                            "# BASELINE config 5 on ONE GPU (2048x1024 panoramas, floor + ceiling, ResNet-152 with 12 input channels, fp16)",
                            f"# bench line: r06_bench_line_config5.json ({lines['config5']['value'] / 1e3:.1f} k hypotheses/s)"], "_config5", 20)
(PRO / "r06_bench_kernel_stats_config5.md").write_text("\n".join(o5) + "\n")

# ---- rasteriser traffic: both FETCH conventions, the benchmark's launch shape and config 5's
def pmc(sub, counter):
    acc = collections.defaultdict(list)
    f = newest(str(SRC / sub / "**" / "*counter_collection.csv"), recursive=True)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "bev_" in r["Kernel_Name"]:
            acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    best = {}
    for (k, g), v in acc.items():            # the largest grid of every kernel = the whole-shard launch
        if k not in best or g > best[k][0]:
            best[k] = (g, sum(v) / len(v))
    return {k: v[1] for k, v in best.items()}

def ras_table(fsub, wsub, n, title, alg):
    fe, wr = pmc(fsub, "FETCH_SIZE"), pmc(wsub, "WRITE_SIZE")
    ks = [k for k in ("bev_splat_kernel<false>", "bev_densify_kernel<false>") if k in fe]
    rows = [(k, fe[k] * 1024, wr[k] * 1024) for k in ks]
    raw = sum(r[1] + r[2] for r in rows) / n
    cor = sum(2 * r[1] + r[2] for r in rows) / n
    txt = title + ["", "| kernel (per launch) | FETCH_SIZE, GB | FETCH_SIZE x 2, GB | WRITE_SIZE, GB |", "|---|---|---|---|"]
    for k, f, w in rows:
        txt.append(f"| `{k}` | {f / 1e9:.2f} | {2 * f / 1e9:.2f} | {w / 1e9:.2f} |")
    txt += [f"| **whole rasteriser** | **{sum(r[1] for r in rows) / 1e9:.2f}** | **{2 * sum(r[1] for r in rows) / 1e9:.2f}** | **{sum(r[2] for r in rows) / 1e9:.2f}** |", "",
            f"Per render: **{raw / 1e6:.2f} MB** (FETCH_SIZE as counted) ... **{cor / 1e6:.2f} MB** (FETCH_SIZE x 2, the guide's gfx950 correction, the convention `roofline.traffic` and",
            f"`roofline_verifier.traffic` share) against {alg / 1e6:.3f} MB algorithmic (SURVEY 8d): {raw / alg:.2f}x ... {cor / alg:.2f}x; writes alone {sum(r[2] for r in rows) / n / 1e6:.2f} MB per render."]
    return txt, raw, cor

t1, raw1, cor1 = ras_table("pmc_fetch", "pmc_write", 4096,
    ["# HBM traffic counters of the rasteriser, round 6 (MI355X, rocprofv3 --pmc, one counter per pass, with --kernel-trace only)", "",
     "Command (per pass): `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/measure/pmc_render.py 4096 64` -- the BENCHMARK's",
     "launch shape: 4096 renders per launch over 64 panoramas of 1024x512 -> 501x501 BEV, issued in panorama order as the pipeline does.  The counters are in KiB.",
     "Since round 6 the densify kernel's last phase also writes the verifier sample (salve_bev_densify_tiles): its 0.80 MB of fp16 NHWC tile per render and the",
     "0.20 MB pretiled second image it reads are IN these counters and NOT in SURVEY 8d's 2.555 MB (with them: 3.56 MB algorithmic per render).",
     "Round 5 (tiles in a launch of their own, not counted here): 2.24 ... 2.72 MB per render.  Round 3: 6.15 ... 8.8 MB."], 2555243)
hit = pmc("pmc_hit", "TCC_HIT_sum"); mis = pmc("pmc_hit", "TCC_MISS_sum")
t1 += ["", "L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS)): " + ", ".join(f"`{k}` {100 * hit[k] / (hit[k] + mis[k]):.0f} %" for k in hit if "splat" in k or "densify" in k) + "."]
t5, raw5, cor5 = ras_table("pmc5_fetch", "pmc5_write", 8192,
    ["", "## BASELINE config 5's launch shape", "",
     "`python3 tools/measure/pmc_render.py 8192 16 1024x2048 floor,ceiling`: 8192 renders per launch (4096 hypotheses x floor + ceiling) over 16 panoramas of 2048x1024."], 7962171)
(PRO / "r06_pmc_traffic.md").write_text("\n".join(t1 + t5) + "\n")
shutil.copy(SRC / "ras_traffic.txt", PRO / "r06_pmc_traffic_raw.txt")
with open(PRO / "r06_pmc_traffic_raw.txt", "a") as f:
    f.write("\n# config 5 launch shape\n" + (SRC / "ras_traffic_config5.txt").read_text())
shutil.copy(SRC / "ras_sq.txt", PRO / "r06_rasteriser_sq_raw.txt")
for k, v in lines.items():
    (PRO / ("r06_bench_line.json" if k == "default" else f"r06_bench_line_{k}.json")).write_text(json.dumps(v) + "\n")

# ---- verifier traffic tables (tools/gpu_r5_traffic.sh -> tools/resnet_traffic_report.py)
t = {"_comment": "Counted HBM traffic (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes; bytes_per_unit_raw_fetch = with FETCH_SIZE as counted) per unit of work, at the launch shape named by the key; written by tools/measure/refresh_profiles_r6.py from gpurun_out/. bench.py reads this file; a workload without an entry reports traffic: null.",
     "rasteriser/1024x512/launch4096": {"bytes_per_unit": cor1, "bytes_per_unit_raw_fetch": raw1, "unit": "render", "source": "profiles/r06_pmc_traffic.md"},
     "rasteriser/2048x1024/launch8192": {"bytes_per_unit": cor5, "bytes_per_unit_raw_fetch": raw5, "unit": "render", "source": "profiles/r06_pmc_traffic.md (config 5)"}}
for layers, ch, name in ((50, 6, "r06_resnet_traffic.md"), (152, 12, "r06_resnet152_traffic.md")):
    TRF = ROOT / "gpurun_out" / f"r6traffic_v{layers}"
    if not (TRF / "summary.json").exists():
        continue
    shutil.copy(TRF / "report.md", PRO / name)
    s = json.load(open(TRF / "summary.json"))
    B = s["batch"]
    t[f"verifier/resnet{layers}-{ch}ch/launch{B}"] = {"bytes_per_unit": (s["counted_read_bytes"] + s["counted_write_bytes"]) / B,
                                                      "bytes_per_unit_raw_fetch": (s["counted_read_bytes"] / 2 + s["counted_write_bytes"]) / B, "unit": "sample", "source": f"profiles/{name}"}
    t[f"verifier_algorithmic/resnet{layers}-{ch}ch"] = {"bytes_per_unit": s["alg_bytes"] / B, "unit": "sample",
                                                       "source": f"profiles/{name} (activations in + residual + out per launch at the present fusion level, fp16, + weights)"}
# ---- the densify kernel's issue counters (SQ pass over tools/pmc_render.py 4096 64): the bound that binds it (bench.py: roofline.valu_busy ...)
sq = {c: pmc("dsq", c).get("bev_densify_kernel<false>") for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")}
if all(v is not None for v in sq.values()):
    t["densify_issue/1024x512/launch4096"] = {
        "valu_busy": round(sq["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sq["SQ_BUSY_CYCLES"] / 32), 4),
        "vector_insts_per_render": round(sq["SQ_INSTS_VALU"] / 4096), "scalar_insts_per_render": round(sq["SQ_INSTS_SALU"] / 4096),
        "unit": "wave-instructions per render; valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines)",
        "source": "profiles/r06_rasteriser_sq_raw.txt"}
json.dump(t, open(PRO / "traffic.json", "w"), indent=1)
print("\n".join(o1[:24])); print("\n".join(t1[-6:] + t5[-3:]))
