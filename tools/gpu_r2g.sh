#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
export SALVE_CONV_WIDE=0
step 300 resnet.log python tools/bench_resnet.py 50 512,668,1000,1024,1336,1366,2048
for b in 1000 1336; do step 120 conv_$b.log python tools/bench_conv.py $b; done
grep -v amdgpu "$OUT/resnet.log"
for b in 1000 1336; do echo "--- B=$b"; grep -v amdgpu "$OUT/conv_$b.log" | cut -c1-64; done
