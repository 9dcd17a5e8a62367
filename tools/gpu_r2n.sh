#!/bin/bash
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 tools/repro/pk_f32_next_to_mfma 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_rasteriser.py -q -x 2>&1 | tail -2
