#!/bin/bash
# Development: A/B of the product library against tools/_abl/libsalve_prev.so on the densify ablation, alternating.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
for r in 1 2; do for L in new prev; do
  if [ $L = prev ]; then export SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_prev.so; else unset SALVE_HIP_LIB; fi
  for sc in box cluttered; do
    step 150 ${L}_${sc}_$r.log python tools/densify_ablation.py 2048 $sc || exit 1
    echo "$L $sc $r: $(grep -h 'per render' $OUT/${L}_${sc}_$r.log | sed 's/.*launch: //')  $(grep -h 'general walk (E2)' $OUT/${L}_${sc}_$r.log | sed 's/.*(E2) *//')  noF $(grep -h 'rasterisation (F)' $OUT/${L}_${sc}_$r.log | sed 's/.*(F) *//')"
  done
done; done
