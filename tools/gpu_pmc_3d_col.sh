#!/bin/bash
# config 4, template kernel: FETCH_SIZE per launch, plane-periodic dealing with / without z-columns
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pmc3d_col}; mkdir -p $OUT; export TMPDIR=/tmp
export LSBENCH_HIP_SPMV_TUNE=70 LSBENCH_HIP_FORCE_PERIOD=1 LSBENCH_HIP_SPMV_GRID=1536
for col in 0 1; do
  if [ $col = 1 ]; then export LSBENCH_HIP_SELL_COL=1; else unset LSBENCH_HIP_SELL_COL; fi
  d=$OUT/col${col}
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$d" -- python3 bench.py --workload lap3d --spmv 5 --fixed-iters 30 --steps 1 --warmup 0 --cpu-seconds 0 --cfg4 0 --general-values 0 > $d.log 2>&1; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
  python3 - "$d" "$d.log" "col=$col" <<'PY'
import csv, sys, glob, json
v = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_spmv_tmpl" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v.append(float(r["Counter_Value"]))
v.sort()
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])["roofline"]
print(sys.argv[3], "launches", len(v), "FETCH x2 = %.1f MB" % (2 * v[len(v) // 2] * 1024 / 1e6), "launch %.1f us b2b %.1f us" % (d["launch_ms"] * 1e3, d["back_to_back_launch_ms"] * 1e3), "period", d["xcd_period_slices"])
PY
  rm -rf "$d"
done
