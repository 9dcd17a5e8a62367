#!/bin/bash
# Round 6, development: per-launch durations of one forward at batch 4096 under rocprofv3 --kernel-trace, for several SALVE_RESNET_FLAGS values.
# usage (GPU box): bash tools/measure/gpu_r6_trace.sh <out dir under gpurun_out> <flags> [<flags> ...]
set -u
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp
for f in "$@"; do
  export SALVE_RESNET_FLAGS=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$f" -- python3 "$ROOT/tools/measure/trace_resnet.py" 4096 > "$OUT/trace_$f.log" 2>&1 || { echo "trace $f failed"; tail -5 "$OUT/trace_$f.log"; exit 1; }
  python3 "$ROOT/tools/measure/trace_resnet_report.py" "$OUT/trace_$f" > "$OUT/launches_$f.txt"
  rm -rf "$OUT/trace_$f"
  head -12 "$OUT/launches_$f.txt"; tail -1 "$OUT/launches_$f.txt"
done
