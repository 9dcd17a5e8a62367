#!/bin/bash
# Development: the whole GPU suite, then one bench line (no CPU baseline).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/quick
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 600 tests.log python -m pytest tests -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
step 300 bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-}
tail -1 "$OUT/bench.log" | cut -c1-200
