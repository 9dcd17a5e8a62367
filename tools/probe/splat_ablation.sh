#!/bin/bash
# Development: timing-only builds of the splat kernel (SPLAT_ABL) into tools/probe/_abl/ (git-ignored; they travel with gpurun).
# The product source carries no timing switch: they are tools/probe/ablations/timing_switches.patch, applied to a temporary copy here.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
WORK=$(mktemp -d /tmp/salve_abl.XXXXXX)
mkdir -p "$WORK/salve_amd" "$WORK/include" "$WORK/tools/probe"
cp -r "$ROOT/salve_amd/csrc" "$WORK/salve_amd/csrc"; cp "$ROOT/include/salve_hip.h" "$WORK/include/"; cp -r "$ROOT/tools/probe/ablations" "$WORK/tools/probe/ablations"
# The patches are records pinned to a base commit (tools/probe/ablations/MANIFEST.json): round 6 changed the kernels under them.  On another tree this stops here;
# to re-run an experiment: `git worktree add /tmp/salve_base $(python3 -c "import json;print(json.load(open('$ROOT/tools/probe/ablations/MANIFEST.json'))['base_commit'])")` and run that tree's scripts.
(cd "$WORK" && patch -p1 -s --dry-run < "$ROOT/tools/probe/ablations/timing_switches.patch" > /dev/null) || { echo "timing_switches.patch does not apply to this tree: see tools/probe/ablations/MANIFEST.json (base commit)"; exit 2; }
(cd "$WORK" && patch -p1 -s < "$ROOT/tools/probe/ablations/timing_switches.patch")
mkdir -p "$ROOT/tools/probe/_abl"
cd "$WORK/salve_amd/csrc"
for abl in ${@:-1 2 4 6 12 28}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSPLAT_ABL=$abl -o $ROOT/tools/probe/_abl/libsalve_splat$abl.so abi.hip bev_render.hip layout.hip resnet.hip &
done
# workgroup sizes (full kernel)
for t in ${SPLAT_T:-}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DSPLAT_THREADS_N=$t -o $ROOT/tools/probe/_abl/libsalve_splatT$t.so abi.hip bev_render.hip layout.hip resnet.hip &
done
wait
ls -la $ROOT/tools/probe/_abl/
