#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r2u
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2u/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2u/tests.log
for c in 1024 2048 4096; do for s in 3 2 1; do
  if [ $s = 1 ]; then extra="--no-overlap"; else extra="--streams $s"; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --chunk $c $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('chunk $c streams $s:', d['value'], 'hyp/s  verifier', d['roofline_verifier']['launch_ms'], 'ms  densify', d['roofline']['densify_ms'], 'scatter', d['roofline']['scatter_ms'])"
done; done
