#!/bin/bash
# Development (CPU box): registers, scratch and LDS of every kernel of one translation unit, from `hipcc -S`.
# usage: tools/measure/isa_regs.sh resnet.hip [name filter]        (writes /tmp/isa/<file>.s)
set -e
src="$(cd "$(dirname "$0")/../.." && pwd)/salve_amd/csrc/$1"
mkdir -p /tmp/isa
out=/tmp/isa/$(basename "$1" .hip).s
flags="-ffp-contract=off"; case "$1" in bev_render.hip|layout.hip) flags="$flags -fno-slp-vectorize";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -S --cuda-device-only "$src" -o "$out" 2>&1 | grep -v "hip-link" || true
awk '/\.name: /{n=$2} /\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.group_segment_fixed_size|\.sgpr_spill_count/{v[$1]=$2} /\.wavefront_size/{print n, "vgpr", v[".vgpr_count:"], "vspill", v[".vgpr_spill_count:"], "sspill", v[".sgpr_spill_count:"], "scratch", v[".private_segment_fixed_size:"], "lds", v[".group_segment_fixed_size:"]}' "$out" | grep -i "${2:-.}" | c++filt | sed 's/(anonymous namespace):://g'
