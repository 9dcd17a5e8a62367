#!/bin/bash
# Development: per-launch durations of one forward at small batches (does a sub-batch whose activations fit the 256 MB
# memory-side cache run the layer-1 / layer-2 launches faster per sample?).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/subbatch
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for B in ${BATCHES:-32 64 128 4096}; do
  timeout -k 10 200 rocprofv3 --kernel-trace -d "$OUT/t$B" -o t --output-format csv -- python3 tools/trace_resnet.py $B > "$OUT/run$B.log" 2>&1 || { echo "trace $B failed"; tail -5 "$OUT/run$B.log"; exit 1; }
  python3 tools/trace_resnet_report.py "$OUT/t$B" > "$OUT/report$B.txt" 2>&1
  rm -rf "$OUT/t$B"
  tail -1 "$OUT/report$B.txt"
done
