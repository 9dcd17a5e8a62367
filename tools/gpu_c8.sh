#!/bin/bash
# Development: the 8-phase convolution kernel -- bit-identity tests, verifier suite, forward time with and without it.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c8
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 500 tests.log python -m pytest tests/test_gpu_conv8.py tests/test_gpu_verifier.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
step 200 auto.log python tools/bench_resnet.py 50 512,2048,4096
SALVE_CONV_WIDE=0 step 200 off.log python tools/bench_resnet.py 50 512,2048,4096
step 200 auto2.log python tools/bench_resnet.py 50 512,2048,4096
echo auto; grep -v amdgpu "$OUT/auto.log"; echo off; grep -v amdgpu "$OUT/off.log"; echo auto; grep -v amdgpu "$OUT/auto2.log"
