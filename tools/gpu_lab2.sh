#!/bin/bash
# gpurun helper: run the SpMV lab on the given problems + one FETCH_SIZE pass.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-lab}; shift
mkdir -p "$OUT"; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "rc=$rc"; if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
step lab_lap2d 300 tools/spmv_lab lap2d 3162 15
step lab_lap3d 300 tools/spmv_lab lap3d 256 9
step pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- tools/spmv_lab lap2d 3162 1
grep -v "^check.*ok" $OUT/lab_lap2d.log; grep -v "^check.*ok" $OUT/lab_lap3d.log
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/pmc_fetch/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:48]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("FETCH_SIZE x2 = %7.1f MB  %s" % (2 * sum(v) / len(v) * 1024 / 1e6, k))
PY
