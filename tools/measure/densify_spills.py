"""Where the SGPR spill reloads of bev_densify_kernel sit (CPU only: hipcc -S, no GPU).

The kernel runs at the limit of the scalar register file (~100 spilled SGPRs, kept in VGPR lanes: v_writelane / v_readlane).  Where
the allocator puts the reloads is decided anew by every change anywhere in the kernel, and when they land inside the row loop of
phase B or the lean-walk loop the kernel issues 5-10 % more vector instructions (round 3: 176 k <-> 291 k per render in phase B
for the same source of that phase).  This prints, for the product instantiation, the reloads inside every loop of 200+ lines;
compare before and after a change.   usage: python tools/measure/densify_spills.py [extra hipcc flags ...]"""
import re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
with tempfile.TemporaryDirectory() as tmp:
    out = Path(tmp) / "bev.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", *sys.argv[1:], "-S",
                    "--cuda-device-only", "-o", str(out), str(ROOT / "salve_amd" / "csrc" / "bev_render.hip")], check=True, stderr=subprocess.DEVNULL)
    text = out.read_text().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_ZN\d+_GLOBAL__N_118bev_densify_kernelILb0E.*:", l))
end = next(i for i, l in enumerate(text) if i > start and ".amdhsa_kernel" in l and "bev_densify_kernelILb0E" in l)
lines = text[start:end]
cnt = lambda a, b, pat: sum(1 for l in lines[a:b] if re.match(r"\s+" + pat, l))
print(f"bev_densify_kernel<false>: {cnt(0, len(lines), 'v_')} vector instructions, {cnt(0, len(lines), 'v_readlane')} v_readlane, "
      f"{cnt(0, len(lines), 'v_writelane')} v_writelane, {cnt(0, len(lines), 'v_mul_lo_u32')} v_mul_lo_u32 (static)")
hdr = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)", l)
    if m:
        hdr[m.group(1)[1:]] = [i, i]
for i, l in enumerate(lines):
    for m in re.finditer(r"Header=(BB\d+_\d+)", l):
        if "L" + m.group(1) in hdr:
            hdr["L" + m.group(1)][1] = max(hdr["L" + m.group(1)][1], i)
for h, (a, b) in sorted(hdr.items(), key=lambda kv: kv[1][0]):
    j = b + 1
    while j < len(lines) and not lines[j].startswith(".LBB"):
        j += 1
    if j - a >= 200:
        print(f"  loop at line {a:5d} ({j - a:4d} lines): {cnt(a, j, 'v_'):4d} vector, {cnt(a, j, 's_'):4d} scalar, {cnt(a, j, 'v_readlane'):3d} spill reloads")
