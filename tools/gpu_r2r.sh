#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for ko in tap chunk; do
  echo "--- K order $ko"
  SALVE_KORDER=$ko timeout -k 10 120 python tools/bench_conv.py 512 2>&1 | grep -v amdgpu | cut -c1-64
  SALVE_KORDER=$ko timeout -k 10 120 python tools/bench_resnet.py 50 512,1024 2>&1 | grep resnet
done
SALVE_KORDER=chunk timeout -k 10 300 python -m pytest tests/test_gpu_verifier.py -q 2>&1 | tail -2
