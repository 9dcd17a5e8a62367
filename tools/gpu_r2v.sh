#!/bin/bash
# A/B: cheaper im2col addressing in conv_igemm (3x3 shapes), previous library beside the new one.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2v
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
export SALVE_BENCH_ONLY=${ONLY:-3x3}
step 200 conv_new.log python tools/bench_conv.py 512
SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_prev.so step 200 conv_prev.log python tools/bench_conv.py 512
unset SALVE_BENCH_ONLY
step 400 tests.log python -m pytest tests/test_gpu_verifier.py -m gpu -q
step 300 bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
cat "$OUT/conv_new.log" "$OUT/conv_prev.log"; tail -3 "$OUT/tests.log"; tail -1 "$OUT/bench.log" | cut -c1-200
