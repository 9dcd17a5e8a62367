#!/bin/bash
# Development (GPU box): bench lines of the product library and of variant builds (tools/probe/build_variant.sh) in alternating runs on ONE box.
# usage: bash tools/probe/ab_lib_bench.sh <out dir under gpurun_out> "<bench args>" <tag> [<tag> ...]     (tag "product" = the in-tree library)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; ARGS=$2; shift 2
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = product ]; then unset SALVE_HIP_LIB; else export SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/_abl/libsalve_$tag.so; fi
    timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-calibration --no-config5 --no-power-probe $ARGS > "$OUT/${tag}_$round.json" 2> "$OUT/${tag}_$round.err" || { echo "$tag failed"; tail -3 "$OUT/${tag}_$round.err"; exit 1; }
    python - "$OUT/${tag}_$round.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(f"{sys.argv[2]:10s} {d['value']:9.0f} hyp/s  scatter {r['scatter_ms']:.3f}  densify {r['densify_ms']:.3f}  verifier {d['roofline_verifier']['launch_ms']:.2f} ms", flush=True)
PY
  done
done
