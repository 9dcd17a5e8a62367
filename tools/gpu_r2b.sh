#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
step 600 tests.log python -m pytest tests -m gpu -x -q -s
step 120 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 300 bench.log python bench.py --steps 5 --warmup 2
step 200 bench_cluttered.log python bench.py --steps 5 --warmup 2 --scene cluttered --no-cpu-baseline
tail -5 "$OUT/tests.log"; tail -2 "$OUT/smoke.log"; tail -1 "$OUT/bench.log"; tail -1 "$OUT/bench_cluttered.log"
