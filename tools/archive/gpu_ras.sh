#!/bin/bash
# Development: rasteriser parity tests, the densify phase ablation on both scenes, then the HBM traffic counters.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ras
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 500 tests.log python -m pytest tests/test_gpu_rasteriser.py tests/test_gpu_fullsize.py tests/test_gpu_ingest.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
step 150 abl_box.log python tools/densify_ablation.py 2048 box || exit 1
step 150 abl_clu.log python tools/densify_ablation.py 2048 cluttered || exit 1
grep -hv amdgpu "$OUT/abl_box.log" "$OUT/abl_clu.log" | grep "per render\|any star"
bash tools/gpu_traffic.sh | grep -A1 "bev_"
