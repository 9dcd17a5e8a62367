"""Per-layer table from a rocprofv3 kernel trace of tools/trace_resnet.py: the LAST forward's launches in launch order,
joined with the ResNet-50 layer shapes (flops, activation + weight bytes)."""
import csv, glob, sys
d = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("conv_igemm", "maxpool", "avgpool"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# (name, M = B*Ho*Wo, K, N, bytes_in, bytes_out_and_res)
L = []
def conv(name, hin, cin, cout, k, s, res=False):
    ho = hin // s
    M = B * ho * ho
    inb = B * hin * hin * cin * 2 if k > 1 or s == 1 else B * ho * ho * cin * 2 * (1 if s == 1 else 1)
    if s == 2 and k == 1: inb = B * hin * hin * cin * 2 // 2   # half of the rows are touched (every other pixel shares 64B lines)
    L.append((name, M, k * k * cin, cout, inb + k * k * cin * cout * 2, M * cout * 2 * (2 if res else 1)))
    return ho
h = conv("conv1 7x7/2", 224, 8, 64, 7, 2)
L.append(("maxpool", B * 56 * 56, 0, 64, B * 112 * 112 * 64 * 2, B * 56 * 56 * 64 * 2)); h = 56
cin = 64
for si, (n, mid) in enumerate(((3, 64), (4, 128), (6, 256), (3, 512))):
    for bi in range(n):
        s = 2 if (bi == 0 and si > 0) else 1
        conv(f"l{si+1}.{bi}.a 1x1", h, cin, mid, 1, 1)
        h2 = conv(f"l{si+1}.{bi}.b 3x3/{s}", h, mid, mid, 3, s)
        if bi == 0: conv(f"l{si+1}.{bi}.ds 1x1/{s}", h, cin, mid * 4, 1, s)
        conv(f"l{si+1}.{bi}.c 1x1+res", h2, mid, mid * 4, 1, 1, res=True)
        h, cin = h2, mid * 4
L.append(("avgpool+fc", B, 0, 0, B * 49 * 2048 * 2, 0))
n_per = len(L)
assert len(rows) % n_per == 0, (len(rows), n_per)
last = rows[-n_per:]
tot = 0
print(f"{'layer':22s} {'M':>8s} {'K':>5s} {'N':>5s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
for (name, M, K, N, bi, bo), r in zip(L, last):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    print(f"{name:22s} {M:8d} {K:5d} {N:5d} {dur:8.1f} {2*M*K*N/dur/1e6:8.0f} {(bi+bo)/dur/1e3:7.0f}")
print("total", round(tot, 1), "us")
