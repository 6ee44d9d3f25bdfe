#!/bin/bash
# two-phase SpMV, phase-1 work item size (entries per workgroup), K = 1
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pl_items}; mkdir -p $OUT; export TMPDIR=/tmp
for it in ${2:-131072 65536 32768 16384}; do
  LSBENCH_HIP_PB_ITEM=$it timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/i$it -- python3 tools/gpu_pl_blocks.py 1 > $OUT/i$it.log 2>&1 || { echo "item=$it failed"; tail -5 $OUT/i$it.log; exit 1; }
  find $OUT -name '*kernel_trace.csv' -delete
  python3 - $OUT/i$it $it <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_pb_" in r["Name"]:
        print("  item=%s %-16s avg %.1f us min %.1f" % (sys.argv[2], r["Name"].split("(")[0][-16:], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
