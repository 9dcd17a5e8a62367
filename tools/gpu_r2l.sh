#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
SALVE_STEM_FUSE=0 timeout -k 10 120 python tools/bench_stem.py 512 2>&1 | grep stem
timeout -k 10 120 python tools/bench_stem.py 512 2>&1 | grep stem
for t in NO_MFMA NO_LOAD NO_EPI NO_POOL; do SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_STEM_$t.so timeout -k 10 120 python tools/bench_stem.py 512 2>&1 | grep stem; done
