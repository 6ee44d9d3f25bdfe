#!/bin/bash
# two-phase SpMV on config 5: per-phase kernel times of the current build (rocprofv3 means over 12 SpMVs)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pl_phases}; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r$rep -- python3 tools/gpu_pl_blocks.py 1 > $OUT/r.log 2>&1 || { echo "failed"; tail -5 $OUT/r.log; exit 1; }
  find $OUT -name '*kernel_trace.csv' -delete
  python3 - $OUT/r$rep <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
t = {}
for r in csv.DictReader(open(f)):
    if "k_pb_" in r["Name"]:
        t["p1" if "products" in r["Name"] else "p2"] = (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3)
print("phase 1 avg %.1f min %.1f us | phase 2 avg %.1f min %.1f us | sum %.1f us" % (t["p1"] + t["p2"] + (t["p1"][0] + t["p2"][0],)))
PY
done
