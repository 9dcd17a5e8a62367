#!/bin/bash
# HBM traffic counters of the rasteriser (FETCH_SIZE / WRITE_SIZE, one counter per rocprofv3 pass, --kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
rm -rf "$OUT"; mkdir -p "$OUT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
cd /tmp
step 200 pmc_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
step 200 pmc_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/pmc_fetch" > "$OUT/traffic.txt"; python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/pmc_write" >> "$OUT/traffic.txt"
cat "$OUT/traffic.txt"
