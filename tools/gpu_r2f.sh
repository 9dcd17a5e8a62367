#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
for v in 0 e; do
  SALVE_CONV_WIDE=$v step 120 conv_$v.log python tools/bench_conv.py 512
  SALVE_CONV_WIDE=$v step 120 conv1k_$v.log python tools/bench_conv.py 1024
  SALVE_CONV_WIDE=$v step 120 resnet_$v.log python tools/bench_resnet.py 50 512,1024
done
for v in 0 e; do echo "--- $v"; grep -v amdgpu "$OUT/conv_$v.log" | cut -c1-64; grep -v amdgpu "$OUT/conv1k_$v.log" | cut -c1-64; grep -v amdgpu "$OUT/resnet_$v.log"; done
