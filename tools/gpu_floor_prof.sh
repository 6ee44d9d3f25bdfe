#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/floorprof; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --workload lap2d:nx=3162,ny=395 --fixed-iters 3000 --steps 2 --warmup 1 --cpu-seconds 0 --krylov ${1:-cg1} > $OUT/run.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/t/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:5]:
    print(r["Name"].split("(")[0][:40], r["Calls"], "avg %.1f us  min %.1f  max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/run.log
find $OUT -name '*kernel_trace.csv' -delete
