#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-lab3d}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 tools/spmv_lab lap3d 400 7 > $OUT/lab.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
grep -vE "^check.*ok" $OUT/lab.log
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- tools/spmv_lab lap3d 400 1 > $OUT/pmc.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/pmc_fetch/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:48]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("FETCH_SIZE x2 = %8.1f MB  %s" % (2 * sum(v) / len(v) * 1024 / 1e6, k))
PY
