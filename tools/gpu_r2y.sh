#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2y
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 400 tests.log python -m pytest tests/test_gpu_rasteriser.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -2 "$OUT/tests.log"
for L in "" G16R2 G8 G32 G64; do
  if [ -n "$L" ]; then export SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_$L.so; fi
  step 150 abl_box_$L.log python tools/densify_ablation.py 2048 box || exit 1
  step 150 abl_clu_$L.log python tools/densify_ablation.py 2048 cluttered || exit 1
  echo "-- ${L:-default G16R4}"; grep -h "per render\|general walk (E2)" "$OUT/abl_box_$L.log" "$OUT/abl_clu_$L.log"
done
