#!/bin/bash
# Development: ablation builds of the library (one -D flag each) into tools/_abl/ (git-ignored *.so; they travel with gpurun).
set -e
cd "$(dirname "$0")/../salve_amd/csrc"
mkdir -p ../../tools/_abl
for tag in NO_MFMA NO_LOADS NO_DSREAD; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DWIDE_$tag -o ../../tools/_abl/libsalve_$tag.so *.hip &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DWIDE_NO_LOADS -DWIDE_NO_DSREAD -o ../../tools/_abl/libsalve_MFMA_ONLY.so *.hip &
for tag in NO_MFMA NO_LOAD NO_EPI NO_POOL; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DSTEM_$tag -o ../../tools/_abl/libsalve_STEM_$tag.so *.hip &
done
# the alternative convolution kernels d / e / f (SALVE_CONV_WIDE), not in the product library
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -o ../../tools/_abl/libsalve_wide.so *.hip &
wait
ls -la ../../tools/_abl/
