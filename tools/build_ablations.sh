#!/bin/bash
# Development: ablation builds of the library (one -D flag each) into tools/_abl/ (git-ignored *.so; they travel with gpurun).
# The timing-only switches (STEM_NO_*, C8_NO_* ...) are tools/ablations/timing_switches.patch, applied to a temporary copy of the sources here:
# the product source carries only the SALVE_BUILD_ABLATIONS includes of the rejected kernels.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WORK=$(mktemp -d /tmp/salve_abl.XXXXXX)
mkdir -p "$WORK/salve_amd" "$WORK/include" "$WORK/tools"
cp -r "$ROOT/salve_amd/csrc" "$WORK/salve_amd/csrc"; cp "$ROOT/include/salve_hip.h" "$WORK/include/"; cp -r "$ROOT/tools/ablations" "$WORK/tools/ablations"
(cd "$WORK" && patch -p1 -s < "$ROOT/tools/ablations/timing_switches.patch")
cd "$WORK/salve_amd/csrc"
mkdir -p "$ROOT/tools/_abl"
for tag in NO_MFMA NO_LOADS NO_DSREAD; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DWIDE_$tag -o $ROOT/tools/_abl/libsalve_$tag.so *.hip &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DWIDE_NO_LOADS -DWIDE_NO_DSREAD -o $ROOT/tools/_abl/libsalve_MFMA_ONLY.so *.hip &
for tag in NO_MFMA NO_LOAD NO_EPI NO_POOL; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DSTEM_$tag -o $ROOT/tools/_abl/libsalve_STEM_$tag.so *.hip &
done
# the alternative convolution kernels d / e / f (SALVE_CONV_WIDE), not in the product library
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -o $ROOT/tools/_abl/libsalve_wide.so *.hip &
wait
ls -la $ROOT/tools/_abl/
