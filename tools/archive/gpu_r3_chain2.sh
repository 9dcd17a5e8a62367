#!/bin/bash
# Round 3: where the expand_chain kernel's step time goes -- timing-only ablations (SALVE_CHAIN_DBG: an ablation build, tools/build_ablations.sh, loaded with SALVE_HIP_LIB) and a deeper weight prefetch.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3chain2
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" >> "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" >> "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return $rc; }
for v in "1 16" "2 16" "4 16" "3 16" "6 16" "5 16"; do
  set -- $v
  cd /tmp
  SALVE_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libsalve_wide.so SALVE_CHAIN_DBG=$1 SALVE_RESNET_FLAGS=$([ "$2" = 16 ] && echo 256 || echo 0) step 200 t_$1_$2.log rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$1_$2" -- python3 "$GRAFT_REPO_ROOT/tools/trace_resnet.py" 4096 "$OUT/ops.json" || exit 1
  echo "dbg=$1 waves=$2"; python3 "$GRAFT_REPO_ROOT/tools/trace_resnet_report.py" "$OUT/trace_$1_$2" | grep -E "expand_chain|total"
done
find "$OUT" -name "*.db" -delete
