#!/bin/bash
# Development: timing-only builds of the splat kernel (bev_splat.h: SPLAT_ABL) into tools/_abl/ (git-ignored; they travel with gpurun).
set -e
cd "$(dirname "$0")/../salve_amd/csrc"
mkdir -p ../../tools/_abl
for abl in ${@:-1 2 4 6 12 28}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSPLAT_ABL=$abl -o ../../tools/_abl/libsalve_splat$abl.so abi.hip bev_render.hip layout.hip resnet.hip &
done
# workgroup sizes (full kernel)
for t in ${SPLAT_T:-}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSPLAT_THREADS_N=$t -o ../../tools/_abl/libsalve_splatT$t.so abi.hip bev_render.hip layout.hip resnet.hip &
done
wait
ls -la ../../tools/_abl/
