#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_verifier.py -q -x 2>&1 | tail -2
SALVE_STEM_FUSE=0 timeout -k 10 120 python tools/bench_stem.py 512 2>&1 | grep stem
timeout -k 10 120 python tools/bench_stem.py 512 2>&1 | grep stem
timeout -k 10 120 python tools/bench_resnet.py 50 512,1024 2>&1 | grep resnet
