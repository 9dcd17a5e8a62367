#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2x
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 500 tests.log python -m pytest tests/test_gpu_rasteriser.py tests/test_gpu_fullsize.py tests/test_gpu_ingest.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
step 300 bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
step 200 walk.log python tools/profile_walk.py
tail -3 "$OUT/tests.log"; tail -1 "$OUT/bench.log" | cut -c1-250; grep -v amdgpu "$OUT/walk.log"
