#!/bin/bash
# Per-launch HBM traffic (FETCH_SIZE / WRITE_SIZE), L2 hit rate and SQ counters of ONE verifier forward, each counter set in a
# rocprofv3 pass of its own (--kernel-trace only, the program directly after --).
#   usage: gpu_verifier_traffic.sh [batch = 4096] [layers = 50] [out = r6traffic_v50]
set -u
export TMPDIR=/tmp
B=${1:-4096}; L=${2:-50}; TAG=${3:-r6traffic_v$L}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
T="$GRAFT_REPO_ROOT/tools/measure/trace_resnet.py"
cd /tmp
step 300 trace.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $T $B "$OUT/ops.json" $L && \
step 300 pmc_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $T $B "$OUT/ops.json" $L && \
step 300 pmc_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $T $B "$OUT/ops.json" $L && \
step 300 pmc_l2.log rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- python3 $T $B "$OUT/ops.json" $L && \
step 300 pmc_sq.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq" -- python3 $T $B "$OUT/ops.json" $L && \
step 300 pmc_clk.log rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_clk" -- python3 $T $B "$OUT/ops.json" $L
find "$OUT" -name "*.db" -delete
cat "$OUT/steps.log"
python3 "$GRAFT_REPO_ROOT/tools/measure/resnet_traffic_report.py" "$OUT" $B "round 6, MI355X, ResNet-$L" > "$OUT/report.md" 2> "$OUT/report.err"; tail -60 "$OUT/report.md" | cut -c1-220; tail -5 "$OUT/report.err"
