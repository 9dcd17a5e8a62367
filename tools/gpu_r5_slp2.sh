#!/bin/bash
# Round 5: the SLP-vectorised build WITHOUT the contender guard of sd_share_best (the vertex fence of RasterEmit stays): does the fence
# fire?  Then one bench line with the SLP build (is it any faster?).  The no-guard probe runs last.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5slp2
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 200 probe_product.log python tools/slp_probe.py --save "$OUT/product.npz" || { tail -20 "$OUT/probe_product.log"; exit 1; }
step 300 bench_product.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
SALVE_HIP_LIB=tools/_abl/libsalve_slp.so step 300 bench_slp.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
for f in product slp; do python - "$OUT/bench_$f.log" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[1].split("/")[-1], d["value"], "hyp/s  scatter", d["roofline"]["scatter_ms"], "densify", d["roofline"]["densify_ms"], "verifier", d["roofline_verifier"]["launch_ms"])
PY
done
SALVE_HIP_LIB=tools/_abl/libsalve_slp_noguard.so step 120 probe_noguard.log python tools/slp_probe.py --compare "$OUT/product.npz"
echo "no-guard probe rc=$?"; tail -12 "$OUT/probe_noguard.log"
rm -f "$OUT/product.npz"
