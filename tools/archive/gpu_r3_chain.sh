#!/bin/bash
# Round 3: the expand_chain kernel -- bit-identity tests, then the forward at batch 4096 with the chain kernel on / unsplit / off (SALVE_RESNET_FLAGS 0 / 512 / 64)
# (alternating runs on one box), then a per-launch trace of the default.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3chain
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 400 tests.log python -m pytest tests/test_gpu_verifier.py tests/test_gpu_conv8.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
for i in 1 2; do
  SALVE_RESNET_FLAGS=0 step 200 c2_$i.log python tools/bench_resnet.py 50 4096 && SALVE_RESNET_FLAGS=512 step 200 c1_$i.log python tools/bench_resnet.py 50 4096 && SALVE_RESNET_FLAGS=64 step 200 c0_$i.log python tools/bench_resnet.py 50 4096 || exit 1
done
for f in c2_1 c1_1 c0_1 c2_2 c1_2 c0_2; do echo $f; grep -v amdgpu "$OUT/$f.log"; done
cd /tmp
step 300 trace.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 4096 "$OUT/ops.json"
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/trace_resnet_report.py" "$OUT/trace"
