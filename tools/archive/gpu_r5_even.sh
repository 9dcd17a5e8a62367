#!/bin/bash
# Round 5: per-kernel times of the ResNet-50 forward at batch 4096 with every pixel of layer 2's last output stored (flags 1024) and
# with the even pixels only (default).  usage (GPU box): bash tools/gpu_r5_even.sh
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/r5c
for F in 1024 0; do
  export SALVE_RESNET_FLAGS=$F
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c/even_$F -o t -- python3 tools/bench_resnet.py 50 4096 > gpurun_out/r5c/even_$F.log 2>&1 || exit 1
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/r5c/even_$F/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("flags $F")
for r in rows[:12]:
    print(f"  {r['Name'][:110]:110s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
