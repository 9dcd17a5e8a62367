#!/bin/bash
# Development: ablation builds of the library (one -D flag each) into tools/probe/_abl/ (git-ignored *.so; they travel with gpurun).
# The timing-only switches (STEM_NO_*, C8_NO_* ...) are tools/probe/ablations/timing_switches.patch, applied to a temporary copy of the sources here:
# the product source carries only the SALVE_BUILD_ABLATIONS includes of the rejected kernels.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
WORK=$(mktemp -d /tmp/salve_abl.XXXXXX)
mkdir -p "$WORK/salve_amd" "$WORK/include" "$WORK/tools/probe"
cp -r "$ROOT/salve_amd/csrc" "$WORK/salve_amd/csrc"; cp "$ROOT/include/salve_hip.h" "$WORK/include/"; cp -r "$ROOT/tools/probe/ablations" "$WORK/tools/probe/ablations"
# The patches are records pinned to a base commit (tools/probe/ablations/MANIFEST.json): round 6 changed the kernels under them.  On another tree this stops here;
# to re-run an experiment: `git worktree add /tmp/salve_base $(python3 -c "import json;print(json.load(open('$ROOT/tools/probe/ablations/MANIFEST.json'))['base_commit'])")` and run that tree's scripts.
(cd "$WORK" && patch -p1 -s --dry-run < "$ROOT/tools/probe/ablations/timing_switches.patch" > /dev/null) || { echo "timing_switches.patch does not apply to this tree: see tools/probe/ablations/MANIFEST.json (base commit)"; exit 2; }
(cd "$WORK" && patch -p1 -s < "$ROOT/tools/probe/ablations/timing_switches.patch")
cd "$WORK/salve_amd/csrc"
mkdir -p "$ROOT/tools/probe/_abl"
for tag in NO_MFMA NO_LOADS NO_DSREAD; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DWIDE_$tag -o $ROOT/tools/probe/_abl/libsalve_$tag.so *.hip &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DWIDE_NO_LOADS -DWIDE_NO_DSREAD -o $ROOT/tools/probe/_abl/libsalve_MFMA_ONLY.so *.hip &
for tag in NO_MFMA NO_LOAD NO_EPI NO_POOL; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -DSTEM_$tag -o $ROOT/tools/probe/_abl/libsalve_STEM_$tag.so *.hip &
done
# the alternative convolution kernels d / e / f (SALVE_CONV_WIDE), not in the product library
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSALVE_BUILD_ABLATIONS -o $ROOT/tools/probe/_abl/libsalve_wide.so *.hip &
wait
ls -la $ROOT/tools/probe/_abl/
