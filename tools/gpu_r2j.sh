#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2j
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
step 600 tests.log python -m pytest tests -m gpu -q -x
tail -3 "$OUT/tests.log"
step 300 bench_1s.log python bench.py --steps 5 --warmup 2 --no-overlap --no-cpu-baseline
step 300 bench.log python bench.py --steps 5 --warmup 2 --no-cpu-baseline
tail -1 "$OUT/bench_1s.log"; tail -1 "$OUT/bench.log"
cd /tmp
step 200 pmc_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
step 200 pmc_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/pmc_fetch" > "$OUT/traffic.txt"; python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/pmc_write" >> "$OUT/traffic.txt"
cat "$OUT/traffic.txt"
find "$OUT" -name "*.db" -delete
