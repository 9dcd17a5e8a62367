#!/bin/bash
# Development: verifier tests, forward time at batch 512 and 2048, per-launch trace.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ver
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 400 tests.log python -m pytest tests/test_gpu_verifier.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
step 200 new.log python tools/bench_resnet.py 50 512,2048
SALVE_RESNET_NO_PROJ_FUSE=1 step 200 old.log python tools/bench_resnet.py 50 512,2048
step 200 new2.log python tools/bench_resnet.py 50 512,2048
echo new; grep -v amdgpu "$OUT/new.log"; echo old; grep -v amdgpu "$OUT/old.log"; echo new; grep -v amdgpu "$OUT/new2.log"
cd /tmp
step 200 trace.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/trace_resnet_report.py" "$OUT/trace" | head -8
