#!/bin/bash
# Round 5: the -m gpu suite, one bench line, then the SLP-vectorised build of bev_render.hip (tools/build_slp.sh) against the product
# build on 256 renders (tools/slp_probe.py).  The SLP step runs LAST and under its own short timeout: that build faulted in rounds 2-4.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5slp
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 900 tests.log python -m pytest tests -m gpu -q -x -s || { tail -40 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
grep -E "head x30|fused layout" "$OUT/tests.log" | cut -c1-300
step 400 bench.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
tail -1 "$OUT/bench.log" | cut -c1-1200
step 200 probe_product.log python tools/slp_probe.py --save "$OUT/product.npz" || { tail -20 "$OUT/probe_product.log"; exit 1; }
tail -3 "$OUT/probe_product.log"
SALVE_HIP_LIB=tools/_abl/libsalve_slp.so step 120 probe_slp.log python tools/slp_probe.py --compare "$OUT/product.npz"
echo "slp probe rc=$?"
tail -25 "$OUT/probe_slp.log"
rm -f "$OUT/product.npz"
