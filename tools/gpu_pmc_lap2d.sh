#!/bin/bash
# config 3: HBM traffic of the SpMV kernel (separate --pmc passes) + per-kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pmc_lap2d}; mkdir -p $OUT; export TMPDIR=/tmp
Q="--cpu-seconds 0 --cfg4 0 --fixed-iters 60 --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py $Q > $OUT/$tag.log 2>&1 || echo "pass $c failed"
  python3 - $OUT/$tag <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if fs:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_spmv_sell16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort()
        print("  %s median %.4g over %d launches" % (k, v[len(v) // 2], len(v)))
PY
  find $OUT -name '*kernel_trace.csv' -delete
done
