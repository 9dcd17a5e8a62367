#!/bin/bash
# Round 3, general walk (E2) of bev_densify_kernel: the rasteriser's -m gpu tests, the walk counters / timers of the profile build
# (tools/profile_walk.py) and the per-phase instruction counters (tools/densify_insts.py) of the product build.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e2
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_rasteriser.py tests/test_gpu_fullsize.py tests/test_gpu_utils.py -m gpu -q -x > "$OUT/tests.log" 2>&1 || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
timeout -k 10 300 python tools/profile_walk.py > "$OUT/walk.log" 2>&1 || { tail -20 "$OUT/walk.log"; exit 1; }
grep -v amdgpu "$OUT/walk.log" | cut -c1-700
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc" -- python3 "$GRAFT_REPO_ROOT/tools/densify_insts.py" 2048 > "$OUT/run.log" 2>&1 || { tail -20 "$OUT/run.log"; exit 1; }
find "$OUT" -name "*.db" -delete
python3 "$GRAFT_REPO_ROOT/tools/densify_insts.py" --report "$OUT/pmc" 2048 | tee "$OUT/report.txt" | cut -c1-160
