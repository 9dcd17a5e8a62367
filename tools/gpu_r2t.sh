#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in 512 1024 1366 2048 4096; do for s in 3 1; do
  if [ $s = 1 ]; then extra="--no-overlap"; else extra=""; fi
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --chunk $c $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('chunk $c streams $s:', d['value'], 'hyp/s  verifier', d['roofline_verifier']['launch_ms'], 'ms  densify', d['roofline']['densify_ms'], 'scatter', d['roofline']['scatter_ms'])"
done; done
