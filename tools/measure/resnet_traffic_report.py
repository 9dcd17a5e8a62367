"""Per-launch roofline table of one verifier forward from the counter passes of tools/measure/gpu_verifier_traffic.sh.

For every launch of the LAST of the three forwards: duration (kernel trace, no counters), algorithmic FLOP and bytes of the
ops it executes (activations in + residual + out, fp16; weights once), the two roofline times (FLOP / 2.5 PFLOP/s dense fp16
peak, bytes / 6.3 TB/s achievable HBM: /opt/skills/guides/MI355X_MICROARCH.md) and which binds, the counted HBM bytes
(FETCH_SIZE x 2: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, same guide; WRITE_SIZE as is; both
in KiB units of 1024 B... the counter reports kilobytes), L2 hit rate and MFMA-busy.
usage: resnet_traffic_report.py <dir> <batch> [title tag]"""
import collections, csv, glob, json, sys
d, B = sys.argv[1], int(sys.argv[2])
TAG = sys.argv[3] if len(sys.argv) > 3 else "round 3, MI355X"
VER = ("conv_igemm", "conv8_kernel", "conv_wide", "conv_pc", "bottleneck", "stem_pool", "maxpool", "avgpool", "expand_chain")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]

def trace(sub):
    f = glob.glob(f"{d}/{sub}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in VER)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // 3
    return [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-n:]]

def counters(sub):
    fs = glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        return None
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ks = [v for v in disp.values() if any(k in v["name"] for k in VER)]
    return ks[-(len(ks) // 3):]

ops = json.load(open(f"{d}/ops.json"))
launches = trace("trace")
fe, wr, l2, sq, clk = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_l2"), counters("pmc_sq"), counters("pmc_clk")

def op_cost(o):
    """(FLOP, activation bytes read, bytes written, weight bytes) per sample of one op"""
    if o["op"] == 0:
        K = o["KH"] * o["KW"] * o["Cin"] + (o["Cin2"] if o["in2_buf"] != -2 else 0)
        fl = 2.0 * o["Ho"] * o["Wo"] * o["Cout"] * K
        rd = o["Hi"] * o["Wi"] * o["Cin"] * 2 + (o["Ho"] * o["Wo"] * o["Cout"] * 2 if o["res_buf"] != -2 else 0)
        if o["in2_buf"] != -2:
            rd += o["Ho"] * o["Wo"] * o["Cin2"] * 2   # the strided pixels only
        return fl, rd, o["Ho"] * o["Wo"] * o["Cout"] * 2, o["Cout"] * K * 2
    if o["op"] == 1:
        return 0.0, o["Hi"] * o["Wi"] * o["Cin"] * 2, o["Ho"] * o["Wo"] * o["Cout"] * 2, 0
    return 2.0 * o["Cin"] * o["Cout"], o["Hi"] * o["Wi"] * o["Cin"] * 2, o["Cout"] * 4, o["Cin"] * o["Cout"] * 4

print(f"# Verifier, one ResNet forward at batch {B}: per-launch rooflines and counted HBM traffic ({TAG})")
print()
print("Durations: rocprofv3 --kernel-trace (no counters), last of three forwards.  alg = algorithmic: FLOP of the ops a launch executes; bytes =")
print("activations read (input, residual, second source) + written, fp16, + the weights once.  t_mfma = FLOP / 2.5 PFLOP/s, t_hbm = bytes / 6.3 TB/s")
print("(achievable HBM rate of the guide); bound = the larger; x = duration / bound time.  counted = FETCH_SIZE x 2 (gfx950 correction of the guide for wide")
print("coalesced reads; the counter is in KB) + WRITE_SIZE, separate --pmc passes.  L2 = TCC_HIT / (TCC_HIT + TCC_MISS).  MFMA = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles).")
print()
print("| # | kernel | ops | us | alg GFLOP | alg MB | t_mfma us | t_hbm us | bound | x | TFLOP/s | counted rd MB | counted wr MB | counted/alg | GB/s counted | L2 hit | MFMA busy |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
oi = 0
tot = collections.Counter()
for i, (name, us) in enumerate(launches):
    nxt = "bottleneck" in name and name.rstrip().endswith("false, true>")   # the NEXT form: the block + the following block's first 1x1 convolution
    take = 4 if nxt else 3 if "bottleneck" in name else (2 if ("stem_pool" in name or ("expand_chain" in name and ", true" in name)) else 1)
    mine = ops[oi:oi + take]; oi += take
    fl = sum(op_cost(o)[0] for o in mine) * B
    if take == 4:    # NEXT form: input, a quarter of Y (even pixels: its one other reader is a stride-2 shortcut), t1' out
        a, c, n_ = mine[0], mine[2], mine[3]
        by = (a["Hi"] * a["Wi"] * a["Cin"] * 2 + (c["Ho"] // 2) * (c["Wo"] // 2) * c["Cout"] * 2 + n_["Ho"] * n_["Wo"] * n_["Cout"] * 2) * B
    elif take == 3:    # fused block: input + residual (the same tensor: once) + output
        a, c = mine[0], mine[2]
        by = (a["Hi"] * a["Wi"] * a["Cin"] * 2 + c["Ho"] * c["Wo"] * c["Cout"] * 2) * B
    elif take == 2 and "expand_chain" in name:  # expand + residual + next reduce: t2 and X in, Y and t1' out (Y is not read back)
        c, a = mine
        by = (c["Hi"] * c["Wi"] * c["Cin"] * 2 + 2 * c["Ho"] * c["Wo"] * c["Cout"] * 2 + a["Ho"] * a["Wo"] * a["Cout"] * 2) * B
    elif take == 2:  # stem + pool: input + pooled output
        by = (mine[0]["Hi"] * mine[0]["Wi"] * mine[0]["Cin"] * 2 + mine[1]["Ho"] * mine[1]["Wo"] * mine[1]["Cout"] * 2) * B
    else:
        by = (op_cost(mine[0])[1] + op_cost(mine[0])[2]) * B
    by += sum(op_cost(o)[3] for o in mine)
    tm, th = fl / 2.5e15 * 1e6, by / 6.3e12 * 1e6
    bound = "mfma" if tm > th else "hbm"
    rd = fe[i]["FETCH_SIZE"] * 1024 * 2 if fe else float("nan")
    ww = wr[i]["WRITE_SIZE"] * 1024 if wr else float("nan")
    hit = l2[i]["TCC_HIT_sum"] / max(1.0, l2[i]["TCC_HIT_sum"] + l2[i]["TCC_MISS_sum"]) if l2 else float("nan")
    mf = float("nan")
    if sq and clk:
        cyc = clk[i]["GRBM_GUI_ACTIVE"] / 8.0
        mf = sq[i]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc) if cyc else 0.0
    kinds = "+".join(f"{o['KH']}x{o['KW']}s{o['stride']} {o['Cin']}->{o['Cout']}@{o['Ho']}" if o["op"] == 0 else ("pool" if o["op"] == 1 else "fc") for o in mine)
    print(f"| {i} | {name[:40]} | {kinds} | {us:.0f} | {fl / 1e9:.0f} | {by / 1e6:.0f} | {tm:.0f} | {th:.0f} | {bound} | {us / max(tm, th):.2f} | {fl / us / 1e6:.0f} | {rd / 1e6:.0f} | {ww / 1e6:.0f} | {(rd + ww) / by:.2f} | {(rd + ww) / us / 1e3:.0f} | {100 * hit:.0f} % | {100 * mf:.0f} % |")
    tot["us"] += us; tot["fl"] += fl; tot["by"] += by; tot["rd"] += rd; tot["wr"] += ww; tot["bound"] += max(tm, th); tot["tm"] += tm; tot["th"] += th
print()
print(f"Forward: {tot['us'] / 1e3:.2f} ms; algorithmic {tot['fl'] / 1e12:.2f} TFLOP ({tot['fl'] / tot['us'] / 1e6:.0f} TFLOP/s = {tot['fl'] / tot['us'] / 1e6 / 25:.1f} % of 2.5 PFLOP/s) and {tot['by'] / 1e9:.1f} GB "
      f"({tot['by'] / B / 1e6:.1f} MB per sample); sum of per-launch bound times {tot['bound'] / 1e3:.2f} ms (MFMA alone {tot['tm'] / 1e3:.2f} ms, HBM alone {tot['th'] / 1e3:.2f} ms); "
      f"counted HBM traffic {(tot['rd'] + tot['wr']) / 1e9:.1f} GB = {(tot['rd'] + tot['wr']) / B / 1e6:.1f} MB per sample (reads {tot['rd'] / 1e9:.1f}, writes {tot['wr'] / 1e9:.1f}) = {(tot['rd'] + tot['wr']) / tot['us'] / 1e6:.2f} TB/s average.")
json.dump({"batch": B, "forward_ms": tot["us"] / 1e3, "alg_bytes": tot["by"], "counted_read_bytes": tot["rd"], "counted_write_bytes": tot["wr"],
           "bound_hbm_ms": tot["th"] / 1e3, "bound_mfma_ms": tot["tm"] / 1e3, "sum_of_bounds_ms": tot["bound"] / 1e3}, open(f"{d}/summary.json", "w"))
