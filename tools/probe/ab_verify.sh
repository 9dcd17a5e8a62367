#!/bin/bash
# Development: the verifier's bit-identity / parity tests through the library given as $1, then the forward A/B (tools/probe/ab_forward.sh).
cd $GRAFT_REPO_ROOT
SALVE_HIP_LIB=$1 timeout -k 10 600 python -m pytest tests/test_gpu_verifier.py tests/test_gpu_conv8.py -m gpu -q -x 2>&1 | tail -3
bash tools/probe/ab_forward.sh "$@"
