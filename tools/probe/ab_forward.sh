#!/bin/bash
# Development: the ResNet-50 forward at batch 4096 with the product library and with each library given on the command line, in
# alternating runs on one box (SALVE_HIP_LIB selects the build).   usage: ab_forward.sh <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout -k 10 200 python tools/measure/bench_resnet.py 50 4096 2>&1 | grep "B=4096" | sed 's/^/product: /'
  for V in "$@"; do
    SALVE_HIP_LIB=$V timeout -k 10 200 python tools/measure/bench_resnet.py 50 4096 2>&1 | grep "B=4096" | sed "s#^#$(basename $V): #"
  done
done
