#!/bin/bash
# Round 5: K order of the 3 x 3 convolutions -- forward time (alternating in one process) and FETCH_SIZE per launch in both orders.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c && mkdir -p $OUT
timeout -k 10 300 python3 tools/ab_korder.py 50 4096 > $OUT/korder_ab.log 2>&1 || exit 1
cat $OUT/korder_ab.log
cd /tmp
for O in tap chunk; do
  export SALVE_K_ORDER=$O
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/korder_$O -- python3 $GRAFT_REPO_ROOT/tools/trace_resnet.py 4096 > $OUT/korder_$O.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/korder_$O/**/*counter_collection.csv", recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    e = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    e[1] += float(r["Counter_Value"])
rows = [v for v in disp.values() if "anonymous" in v[0]]
n = len(rows) // 3
print("order $O: FETCH_SIZE x 2 KB -> MB per launch of the last forward")
tot = 0
for i, (name, val, ns) in enumerate(rows[-n:]):
    mb = val * 2 * 1024 / 1e6
    tot += mb
    print(f"  {i:2d} {name[27:90]:64s} {mb:9.0f} MB {ns/1e3:9.0f} us")
print(f"  total {tot:.0f} MB")
PY
done
find $OUT -name "*.db" -delete; rm -rf $OUT/korder_tap $OUT/korder_chunk
