#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r2q
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2q/tests.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r2q/tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
