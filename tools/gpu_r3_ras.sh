#!/bin/bash
# Round 3: rasteriser with the XCD-grouped workgroup order (SALVE_RAS_XCD=1, default) against the natural order (0): parity tests,
# then alternating bench runs on one box.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ras
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
step 600 tests.log python -m pytest tests/test_gpu_ingest.py tests/test_gpu_dataset.py -m gpu -q -x || { tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
for i in 1 2; do
  SALVE_RAS_XCD=2 step 200 new$i.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline && SALVE_RAS_XCD=0 step 200 old$i.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline || exit 1
done
for f in new1 old1 new2 old2; do echo $f; grep '^{' "$OUT/$f.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'scatter', d['roofline']['scatter_ms'], 'densify', d['roofline']['densify_ms'], 'verifier', d['roofline_verifier']['launch_ms'])"; done
