#!/bin/bash
# PMC traffic of the solve's kernels on the 64 M-row 7-point operator (own runs per counter).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc3d; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 bench.py --workload lap3d --fixed-iters 30 --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/$c.log 2>&1 || exit 1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(sys.argv[1] + "/%s/*/*counter_collection.csv" % c))[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v.sort(); res[k][c] = (v[len(v) // 2], len(v))
for k, d in sorted(res.items()):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fb, wb = 2 * d["FETCH_SIZE"][0] * 1024, d["WRITE_SIZE"][0] * 1024
        print("%-40s launches %4d fetch(x2) %.1f MB write %.1f MB total %.1f MB" % (k[:40], d["FETCH_SIZE"][1], fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
PY
find $OUT -name '*kernel_trace.csv' -delete
