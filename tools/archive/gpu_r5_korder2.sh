#!/bin/bash
# Round 5: the 3 x 3 shapes at batch 4096 one by one, K ordered (tap, channel) and (64-channel chunk, tap, channel): time, board power, clock.
cd $GRAFT_REPO_ROOT && OUT=gpurun_out/r5c && mkdir -p $OUT
export SALVE_BENCH_ONLY="3x3" SALVE_BENCH_POWER=1 SALVE_BENCH_REPS=40
for O in tap chunk tap chunk; do
  echo "== K order: $O"
  SALVE_K_ORDER=$O timeout -k 10 200 python3 tools/bench_conv.py 4096 2>&1 | grep -v amdgpu.ids || exit 1
done
