#!/bin/bash
# SQ counters of the rasteriser kernels (two rocprofv3 --pmc passes of 8 counters, --kernel-trace only) on tools/pmc_render.py.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/dsq
rm -rf "$OUT"; mkdir -p "$OUT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
cd /tmp
step 200 p1.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/p1" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
step 200 p2.log rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/p2" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
step 200 p3.log rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p3" -- python3 "$GRAFT_REPO_ROOT/tools/pmc_render.py"
find "$OUT" -name "*.db" -delete
for p in p1 p2 p3; do python3 "$GRAFT_REPO_ROOT/tools/pmc_report.py" "$OUT/$p" bev_; done > "$OUT/report.txt"
cat "$OUT/report.txt"
