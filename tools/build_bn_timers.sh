#!/bin/bash
# Development: timing-only builds of bottleneck_kernel into tools/_abl/ (none of them is the product):
#   libsalve_bn_timers.so        -DSALVE_BN_TIMERS: phase timers, read by tools/bn_phase_time.py
#   libsalve_bn_abl{1,2,3}.so    -DSALVE_BN_ABL: no output stores / every X row reads the zero page / both
#   libsalve_bn_1wg[_ablN].so    -DSALVE_BN_PAD_LDS=24576: one workgroup per CU instead of two (alone and with the switches above)
#   libsalve_sp{1,2,3,4}.so      -DSALVE_STORE_POLICY: the fused block's output stores sc1 / nt / sc0 sc1 / sc1 nt
# Time them with `SALVE_HIP_LIB=tools/_abl/<lib> python tools/bench_resnet.py 50 4096` (the forward's difference is the three launches').
# The product source carries no timing switch: they are tools/ablations/timing_switches.patch, applied to a temporary copy here.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WORK=$(mktemp -d /tmp/salve_abl.XXXXXX)
mkdir -p "$WORK/salve_amd" "$WORK/include" "$WORK/tools"
cp -r "$ROOT/salve_amd/csrc" "$WORK/salve_amd/csrc"; cp "$ROOT/include/salve_hip.h" "$WORK/include/"; cp -r "$ROOT/tools/ablations" "$WORK/tools/ablations"
(cd "$WORK" && patch -p1 -s < "$ROOT/tools/ablations/timing_switches.patch")
cd "$WORK/salve_amd/csrc"
mkdir -p "$ROOT/tools/_abl" /tmp/bn_obj
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
for f in bev_render layout; do hipcc $F -fno-slp-vectorize -c $f.hip -o /tmp/bn_obj/$f.o & done
hipcc $F -c abi.hip -o /tmp/bn_obj/abi.o &
wait
build() { local tag=$1; shift; hipcc $F "$@" -c resnet.hip -o /tmp/bn_obj/r_$tag.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_abl/libsalve_$tag.so /tmp/bn_obj/r_$tag.o /tmp/bn_obj/bev_render.o /tmp/bn_obj/layout.o /tmp/bn_obj/abi.o; }
build bn_timers -DSALVE_BN_TIMERS &
for a in 1 2 3; do build bn_abl$a -DSALVE_BN_ABL=$a & done
wait
build bn_1wg -DSALVE_BN_PAD_LDS=24576 &
for a in 1 2 3; do build bn_1wg_abl$a -DSALVE_BN_PAD_LDS=24576 -DSALVE_BN_ABL=$a & done
wait
for pol in 1 2 3 4; do build sp$pol -DSALVE_STORE_POLICY=$pol & done
wait
ls -la $ROOT/tools/_abl/
