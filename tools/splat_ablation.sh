#!/bin/bash
# Development: timing-only builds of the splat kernel (SPLAT_ABL) into tools/_abl/ (git-ignored; they travel with gpurun).
# The product source carries no timing switch: they are tools/ablations/timing_switches.patch, applied to a temporary copy here.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WORK=$(mktemp -d /tmp/salve_abl.XXXXXX)
mkdir -p "$WORK/salve_amd" "$WORK/include" "$WORK/tools"
cp -r "$ROOT/salve_amd/csrc" "$WORK/salve_amd/csrc"; cp "$ROOT/include/salve_hip.h" "$WORK/include/"; cp -r "$ROOT/tools/ablations" "$WORK/tools/ablations"
(cd "$WORK" && patch -p1 -s < "$ROOT/tools/ablations/timing_switches.patch")
mkdir -p "$ROOT/tools/_abl"
cd "$WORK/salve_amd/csrc"
for abl in ${@:-1 2 4 6 12 28}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSPLAT_ABL=$abl -o $ROOT/tools/_abl/libsalve_splat$abl.so abi.hip bev_render.hip layout.hip resnet.hip &
done
# workgroup sizes (full kernel)
for t in ${SPLAT_T:-}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSPLAT_THREADS_N=$t -o $ROOT/tools/_abl/libsalve_splatT$t.so abi.hip bev_render.hip layout.hip resnet.hip &
done
wait
ls -la $ROOT/tools/_abl/
