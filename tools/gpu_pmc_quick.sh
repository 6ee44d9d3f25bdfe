#!/bin/bash
# one PMC pass (FETCH_SIZE, WRITE_SIZE in separate runs) of a bench.py workload: per-kernel medians
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pmcq}; mkdir -p $OUT; export TMPDIR=/tmp
WL=${2:-lap3d}; IT=${3:-40}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- python3 bench.py --workload $WL --fixed-iters $IT --steps 1 --warmup 0 --cpu-seconds 0 --cfg4 0 --general-values 0 ${@:4} > $OUT/$c.log 2>&1; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if len(d.get("FETCH_SIZE", [])) < 20:
        continue
    f = sorted(d["FETCH_SIZE"]); w = sorted(d.get("WRITE_SIZE", [0]))
    print("%-60s n=%4d FETCH x2 = %8.1f MB  WRITE = %8.1f MB" % (k, len(f), 2 * f[len(f) // 2] * 1024 / 1e6, w[len(w) // 2] * 1024 / 1e6))
PY
find "$OUT" -name '*kernel_trace.csv' -delete 2>/dev/null; find "$OUT" -name '*counter_collection.csv' -size +8M -delete 2>/dev/null
