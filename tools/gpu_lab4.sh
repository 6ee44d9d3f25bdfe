#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-lab4}; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "lap3d 400 7" "lap2d 3162 11" "lap3d 256 7"; do
  timeout -k 10 400 tools/spmv_lab $cfg > $OUT/lab.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
  grep -E "^lap|bandwidth|library|PERIOD|prefetch cap2048|MISMATCH" $OUT/lab.log
done
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- tools/spmv_lab lap3d 400 1 > $OUT/pmc.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/pmc_fetch/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:48]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "cyc" in k: print("FETCH_SIZE x2 = %8.1f MB  %s" % (2 * sum(v) / len(v) * 1024 / 1e6, k))
PY
