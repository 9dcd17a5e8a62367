#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$1
for i in 1 2 3; do
  timeout -k 10 200 python tools/bench_resnet.py 50 4096 2>&1 | grep "B=4096" | sed 's/^/product: /'
  SALVE_HIP_LIB=$V timeout -k 10 200 python tools/bench_resnet.py 50 4096 2>&1 | grep "B=4096" | sed 's/^/variant: /'
done
