#!/bin/bash
# Development: ablation builds of the 8-phase kernel (tools/_abl/libsalve_C8_<tag>.so) on the compute-bound shapes.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c8abl
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
export SALVE_BENCH_ONLY=${ONLY:-l4.conv2,l3.conv1,l4.conv1,l3.conv2}
for L in base full ${VARIANTS}; do
  unset SALVE_HIP_LIB SALVE_RESNET_FLAGS
  if [ $L = base ]; then export SALVE_RESNET_FLAGS=1; else export SALVE_RESNET_FLAGS=2; fi
  if [ $L != full ] && [ $L != base ]; then export SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_C8_$L.so; fi
  step 200 $L.log python tools/bench_conv.py 4096
  echo "-- $L"; grep -v amdgpu "$OUT/$L.log" | head -5
done
