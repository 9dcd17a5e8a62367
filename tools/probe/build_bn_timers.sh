#!/bin/bash
# Development: timing-only builds of bottleneck_kernel into tools/probe/_abl/ (none of them is the product):
#   libsalve_bn_timers.so        -DSALVE_BN_TIMERS: phase timers, read by tools/probe/bn_phase_time.py
#   libsalve_bn_abl{1,2,3}.so    -DSALVE_BN_ABL: no output stores / every X row reads the zero page / both
#   libsalve_bn_1wg[_ablN].so    -DSALVE_BN_PAD_LDS=24576: one workgroup per CU instead of two (alone and with the switches above)
#   libsalve_sp{1,2,3,4}.so      -DSALVE_STORE_POLICY: the fused block's output stores sc1 / nt / sc0 sc1 / sc1 nt
# Time them with `SALVE_HIP_LIB=tools/probe/_abl/<lib> python tools/measure/bench_resnet.py 50 4096` (the forward's difference is the three launches').
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
cd "$WORK/salve_amd/csrc"
mkdir -p "$ROOT/tools/probe/_abl" /tmp/bn_obj
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
for f in bev_render layout; do hipcc $F -fno-slp-vectorize -c $f.hip -o /tmp/bn_obj/$f.o & done
hipcc $F -c abi.hip -o /tmp/bn_obj/abi.o &
wait
build() { local tag=$1; shift; hipcc $F "$@" -c resnet.hip -o /tmp/bn_obj/r_$tag.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/_abl/libsalve_$tag.so /tmp/bn_obj/r_$tag.o /tmp/bn_obj/bev_render.o /tmp/bn_obj/layout.o /tmp/bn_obj/abi.o; }
build bn_timers -DSALVE_BN_TIMERS &
for a in 1 2 3; do build bn_abl$a -DSALVE_BN_ABL=$a & done
wait
build bn_1wg -DSALVE_BN_PAD_LDS=24576 &
for a in 1 2 3; do build bn_1wg_abl$a -DSALVE_BN_PAD_LDS=24576 -DSALVE_BN_ABL=$a & done
wait
for pol in 1 2 3 4; do build sp$pol -DSALVE_STORE_POLICY=$pol & done
wait
ls -la $ROOT/tools/probe/_abl/
