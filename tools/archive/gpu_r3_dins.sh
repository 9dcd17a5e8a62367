#!/bin/bash
# Round 3: instruction counters of bev_densify_kernel per phase (tools/densify_insts.py), product build.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3dins
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc" -- python3 "$GRAFT_REPO_ROOT/tools/densify_insts.py" 2048 > "$OUT/run.log" 2>&1 || { tail -20 "$OUT/run.log"; exit 1; }
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/densify_insts.py" --report "$OUT/pmc" 2048 | tee "$OUT/report.txt"
