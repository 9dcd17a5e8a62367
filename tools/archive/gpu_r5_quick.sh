#!/bin/bash
# Round 5, development: the whole GPU suite (-s: the tests print their measured errors), then one bench line.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5quick
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 900 tests.log python -m pytest tests -m gpu -q -x -s ${TEST_ARGS:-} || { tail -40 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
grep -E "config 4|config 5|launch shape" "$OUT/tests.log" | cut -c1-300
step 400 bench.log python bench.py --steps 10 --warmup 3 ${BENCH_ARGS:-}
tail -1 "$OUT/bench.log" | cut -c1-1500
