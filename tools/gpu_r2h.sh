#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2h
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
export SALVE_BENCH_ONLY="l3.conv2,l4.conv2,l3.conv1,l2.conv2"
for cfg in d e a; do
  for lib in FULL NO_MFMA NO_LOADS NO_DSREAD MFMA_ONLY; do
    if [ $lib = FULL ]; then unset SALVE_HIP_LIB; else export SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_$lib.so; fi
    SALVE_CONV_WIDE=$cfg step 120 abl_${cfg}_$lib.log python tools/bench_conv.py 512
    echo "--- cfg $cfg $lib"; grep -v amdgpu "$OUT/abl_${cfg}_$lib.log" | cut -c1-62
  done
done
