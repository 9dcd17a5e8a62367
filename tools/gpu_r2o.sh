#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r2o
cd "$GRAFT_REPO_ROOT"
export SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_SLP.so
timeout -k 10 600 python -m pytest tests/test_gpu_rasteriser.py -q -x > gpurun_out/r2o/t1.log 2>&1
echo "rc=$?"; head -40 gpurun_out/r2o/t1.log
