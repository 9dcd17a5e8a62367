#!/bin/bash
# Round-2 first GPU probe: baseline tests, float-parity measurements, per-layer verifier trace, SQ counters.
# Every step is bounded by `timeout -k`; a step that times out (124 / 137) ends the script.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() {  # step <seconds> <log> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* " | tee -a "$OUT/steps.log"
    timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1
    local rc=$?
    echo "   rc=$rc" | tee -a "$OUT/steps.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a "$OUT/steps.log"; exit 1; fi
    return 0
}
step 300 tests.log python -m pytest tests -m gpu -x -q
step 200 float_parity.log python tools/probe_float_parity.py
step 120 bench_resnet.log python tools/bench_resnet.py 50 256,512,1024
rocprofv3 -L > "$OUT/counters.txt" 2>&1
cd /tmp
step 200 trace.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
python3 "$GRAFT_REPO_ROOT/tools/trace_resnet_report.py" "$OUT/trace" > "$OUT/trace_report.txt" 2>&1
step 200 pmc1.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc1" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
step 200 pmc2.log rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc2" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
step 200 pmc3.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc3" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
step 200 pmc4.log rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc4" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
# keep the CSVs small enough to merge back: only counter_collection / kernel_trace files
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT" | tee -a "$OUT/steps.log"
echo DONE | tee -a "$OUT/steps.log"
