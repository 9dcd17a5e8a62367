#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
for v in 0 a b c d; do
  SALVE_CONV_WIDE=$v step 120 conv_$v.log python tools/bench_conv.py 512
done
for v in 0 1; do
  SALVE_CONV_WIDE=$v step 120 resnet_$v.log python tools/bench_resnet.py 50 512,1024
done
SALVE_CONV_WIDE=1 step 300 tests.log python -m pytest tests/test_gpu_verifier.py -q -s
paste -d'|' "$OUT"/conv_0.log "$OUT"/conv_a.log | cut -c1-200 | head -3
for v in 0 a b c d; do echo "--- $v"; cat "$OUT/conv_$v.log"; done
cat "$OUT"/resnet_*.log; tail -5 "$OUT/tests.log"
