#!/bin/bash
# Development: the 8-phase convolution kernels -- bit-identity tests, per-shape timing (0 = conv_igemm, 8 = conv8, 9 = conv8b).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c8
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 500 tests.log python -m pytest tests/test_gpu_conv8.py tests/test_gpu_verifier.py -m gpu -q -x -k "eight or alternative" || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
export SALVE_BENCH_ONLY=${ONLY:-l4.conv2,l3.conv1,l4.conv1,l3.conv2,l4.conv3}
for m in 1 2; do
  SALVE_RESNET_FLAGS=$m step 200 m$m.log python tools/bench_conv.py ${BATCH:-4096}
  echo "-- SALVE_RESNET_FLAGS=$m (1 = conv_igemm_kernel only, 2 = the 8-phase kernel wherever it fits)"; grep -v amdgpu "$OUT/m$m.log"
done
