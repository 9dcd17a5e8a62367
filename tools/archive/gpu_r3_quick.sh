#!/bin/bash
# Round 3: the whole -m gpu suite, then bench lines (default; config 5 on one GPU), then the rasteriser traffic counters at the
# benchmark's launch shape (4096 renders per launch).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3quick
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 900 tests.log python -m pytest tests -m gpu -q -s || tail -40 "$OUT/tests.log"
grep -E "passed|failed|differ|dlogit|benchmark launch shape|RCCL|hard-site" "$OUT/tests.log" | tail -12
step 400 bench.log python bench.py --steps 10 --warmup 3 || { tail -30 "$OUT/bench.log"; exit 1; }
tail -1 "$OUT/bench.log" | cut -c1-1800
step 400 bench_c5.log python bench.py --pano-hw 1024x2048 --surfaces floor,ceiling --layers 152 --hyps 1024 --panos 16 --chunk 512 --steps 3 --warmup 1 --no-cpu-baseline || { tail -30 "$OUT/bench_c5.log"; exit 1; }
tail -1 "$OUT/bench_c5.log" | cut -c1-1500
cd /tmp
step 300 pmc_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py" 4096 64 && \
step 300 pmc_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py" 4096 64
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/pmc_fetch" > "$OUT/traffic.txt"; python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/pmc_write" >> "$OUT/traffic.txt"
cat "$OUT/traffic.txt"
