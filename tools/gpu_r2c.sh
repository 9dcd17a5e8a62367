#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
step 600 tests.log python -m pytest tests -m gpu -q -s
step 120 smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 300 bench.log python bench.py --steps 5 --warmup 2
step 200 bench_1s.log python bench.py --steps 5 --warmup 2 --no-overlap --no-cpu-baseline
step 200 bench_2s.log python bench.py --steps 5 --warmup 2 --streams 2 --no-cpu-baseline
tail -8 "$OUT/tests.log"; tail -2 "$OUT/smoke.log"; tail -1 "$OUT/bench.log"; tail -1 "$OUT/bench_1s.log"; tail -1 "$OUT/bench_2s.log"
