#!/bin/bash
# Round 5, second development set: the -m gpu suite, the SLP-vectorised build of bev_render.hip through the rasteriser's own tests and
# timed (tools/splat_time.py), bench lines of the three scenes.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 900 tests.log python -m pytest tests -m gpu -q -x -s || { tail -40 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"; grep -E "hard-site share|wide sample, worst" "$OUT/tests.log" | cut -c1-300
step 300 bench_box.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline
step 300 bench_cluttered.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline --scene cluttered
step 300 bench_noisy.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline --scene noisy
for f in box cluttered noisy; do python - "$OUT/bench_$f.log" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[1].split("/")[-1], d["value"], "hyp/s  ms/step", d["ms_per_step"], "scatter", d["roofline"]["scatter_ms"], "densify", d["roofline"]["densify_ms"], "verifier", d["roofline_verifier"]["launch_ms"])
PY
done
step 200 time_product.log python tools/splat_time.py
export SALVE_HIP_LIB=tools/_abl/libsalve_slp.so
step 200 time_slp.log python tools/splat_time.py
step 600 tests_slp.log python -m pytest tests/test_gpu_rasteriser.py tests/test_gpu_utils.py tests/test_gpu_fullsize.py tests/test_gpu_facade.py -m gpu -q -x
echo "SLP build through the rasteriser suites: rc=$?"; tail -2 "$OUT/tests_slp.log"
echo "--- product"; tail -6 "$OUT/time_product.log"; echo "--- slp"; tail -6 "$OUT/time_slp.log"
