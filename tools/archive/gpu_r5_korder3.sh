#!/bin/bash
# Round 5: per-launch durations of one forward at batch 4096 under both K orders (rocprofv3 --kernel-trace, no counters), twice each.
cd /tmp && export TMPDIR=/tmp && OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c && mkdir -p $OUT
for O in tap chunk tap2 chunk2; do
  export SALVE_K_ORDER=${O%2}
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$O -- python3 $GRAFT_REPO_ROOT/tools/trace_resnet.py 4096 > $OUT/kt_$O.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob
cols = {}
for O in ("tap", "chunk", "tap2", "chunk2"):
    f = glob.glob("$OUT/kt_%s/**/*kernel_trace.csv" % O, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "anonymous" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // 3
    cols[O] = [(r["Kernel_Name"][27:80], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-n:]]
    print(O, "forward, sum of launches: %.2f ms" % (sum(v for _, v in cols[O]) / 1e3))
for i in range(len(cols["tap"])):
    print(f"{i:2d} {cols['tap'][i][0]:54s}" + "".join(f" {cols[O][i][1]:8.0f}" for O in cols))
PY
rm -rf $OUT/kt_tap $OUT/kt_chunk $OUT/kt_tap2 $OUT/kt_chunk2
