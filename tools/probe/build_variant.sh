#!/bin/bash
# Development: the library with extra -D flags on ONE translation unit, into tools/probe/_abl/libsalve_<tag>.so, for same-box A/B through
# SALVE_HIP_LIB (salve_amd/_lib.py) -- e.g.   tools/probe/build_variant.sh band bev_render.hip -DSD_BAND_WINDOW      Never the product.
set -e
tag=$1; unit=$2; shift 2
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/salve_amd/csrc"
mkdir -p "$ROOT/tools/probe/_abl" /tmp/variant_obj_$tag
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
for f in bev_render layout resnet abi; do
  extra=""; [ "$f.hip" = "$unit" ] && extra="$*"
  slp="-fno-slp-vectorize"; [ "$f" = resnet ] && slp=""; [ "$f" = abi ] && slp=""
  hipcc $F $slp $extra -c $f.hip -o /tmp/variant_obj_$tag/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probe/_abl/libsalve_$tag.so" /tmp/variant_obj_$tag/*.o
ls -la "$ROOT/tools/probe/_abl/libsalve_$tag.so"
