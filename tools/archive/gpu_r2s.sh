#!/bin/bash
# Round-2 profile set at the benchmark's defaults (whole-shard launches, one stream): bench lines, rocprofv3 kernel stats of
# the same command, the three-stream variant, SQ counters of the verifier.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2s
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
step 600 tests.log python -m pytest tests -m gpu -q -s
step 400 bench.log python bench.py --steps 20 --warmup 5
step 300 bench_3s.log python bench.py --steps 10 --warmup 3 --streams 3 --no-cpu-baseline
step 300 bench_cluttered.log python bench.py --steps 10 --warmup 3 --scene cluttered --no-cpu-baseline
cd /tmp
step 300 prof1.log rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline
step 300 prof3.log rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --streams 3
step 200 sq1.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/sq1" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
step 200 sq2.log rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d "$OUT/sq2" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
step 200 sq3.log rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq3" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
step 200 trace.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 512
find "$OUT" -name "*.db" -delete
grep -E "passed|failed|hard-site" "$OUT/tests.log" | tail -3
tail -1 "$OUT/bench.log" | cut -c1-400; tail -1 "$OUT/bench_3s.log" | cut -c1-120; tail -1 "$OUT/bench_cluttered.log" | cut -c1-120
