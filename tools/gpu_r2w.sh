#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2w
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
export SALVE_BENCH_ONLY=3x3,l2.conv1,l3.conv1
export SALVE_BENCH_REPS=40
A=$GRAFT_REPO_ROOT/tools/_abl
step 200 swave_utap.log python tools/bench_conv.py 512
SALVE_CONV_NO_UTAP=1 step 200 swave_tab.log python tools/bench_conv.py 512
SALVE_HIP_LIB=$A/libsalve_vwave.so step 200 vwave_utap.log python tools/bench_conv.py 512
SALVE_HIP_LIB=$A/libsalve_vwave.so SALVE_CONV_NO_UTAP=1 step 200 vwave_tab.log python tools/bench_conv.py 512
SALVE_HIP_LIB=$A/libsalve_prev.so step 200 prev.log python tools/bench_conv.py 512
step 200 swave_utap2.log python tools/bench_conv.py 512
for f in swave_utap swave_tab vwave_utap vwave_tab prev swave_utap2; do echo "-- $f"; grep -v amdgpu.ids "$OUT/$f.log"; done
