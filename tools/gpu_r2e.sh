#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p "$OUT"
cd /tmp
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
export SALVE_BENCH_ONLY="l3.conv2,l4.conv2,l3.conv1" SALVE_BENCH_REPS=3
for v in 0 a d; do
  export SALVE_CONV_WIDE=$v
  step 200 p1_$v.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/p1_$v" -- python3 "$GRAFT_REPO_ROOT/tools/bench_conv.py" 512
  step 200 p2_$v.log rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d "$OUT/p2_$v" -- python3 "$GRAFT_REPO_ROOT/tools/bench_conv.py" 512
  step 200 p3_$v.log rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p3_$v" -- python3 "$GRAFT_REPO_ROOT/tools/bench_conv.py" 512
done
for v in 0 a d; do for p in 1 2 3; do echo "### $v pass $p"; python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/p${p}_$v" conv_; done; done > "$OUT/report.txt" 2>&1
find "$OUT" -name "*.db" -delete
