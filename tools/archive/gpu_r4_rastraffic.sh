#!/bin/bash
# Round 4: the rasteriser's HBM traffic counters at the benchmark's launch shape (one rocprofv3 --pmc pass per counter, --kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4traffic
rm -rf "$OUT"; mkdir -p "$OUT"
N=${1:-4096}; P=${2:-64}
cd /tmp
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
step 300 pmc_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py" $N $P
step 300 pmc_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py" $N $P
step 300 pmc_h.log rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_hit" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py" $N $P
find "$OUT" -name "*.db" -delete
for p in pmc_fetch pmc_write pmc_hit; do python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/$p" bev_; done > "$OUT/ras_traffic.txt"
cat "$OUT/ras_traffic.txt"
