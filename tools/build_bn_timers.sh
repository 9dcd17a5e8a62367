#!/bin/bash
# Development: the library with bottleneck_kernel's phase timers (-DSALVE_BN_TIMERS, for tools/bn_phase_time.py) and its timing-only
# builds (-DSALVE_BN_ABL=1: no output stores, 2: every X row reads the zero page, 3: both) into tools/_abl/.
set -e
cd "$(dirname "$0")/../salve_amd/csrc"
mkdir -p ../../tools/_abl
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared"
hipcc $F -DSALVE_BN_TIMERS -o ../../tools/_abl/libsalve_bn_timers.so *.hip &
for a in 1 2 3; do hipcc $F -DSALVE_BN_ABL=$a -o ../../tools/_abl/libsalve_bn_abl$a.so *.hip & done
wait
ls -la ../../tools/_abl/
