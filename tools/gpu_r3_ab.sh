#!/bin/bash
# Round 3: verifier tests, then the forward at batch 4096 with the XCD-contiguous tile mapping on / off (alternating runs on one box).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ab
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 500 tests.log python -m pytest tests/test_gpu_verifier.py tests/test_gpu_conv8.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
for i in 1 2; do
  SALVE_XCD_CONTIG=1 step 200 new$i.log python tools/bench_resnet.py 50 4096 && SALVE_XCD_CONTIG=0 step 200 old$i.log python tools/bench_resnet.py 50 4096 || exit 1
done
for f in new1 old1 new2 old2; do echo $f; grep -v amdgpu "$OUT/$f.log"; done
cd /tmp
step 300 trace.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 4096 "$OUT/ops.json"
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/trace_resnet_report.py" "$OUT/trace"
