#!/bin/bash
# binned SpMV on config 5: window widths, then trace + PMC of the default width
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-plw}; mkdir -p $OUT; export TMPDIR=/tmp
for w in 131072 262144 393216 524288; do
  LSBENCH_HIP_PANEL_COLS=$w timeout -k 10 300 python3 bench.py --workload powerlaw --spmv 6 --cpu-seconds 0 > $OUT/w$w.log 2>&1 || exit 1
  echo "width $w: $(grep -o '"ms_per_step": [0-9.]*' $OUT/w$w.log)"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload powerlaw --spmv 6 --cpu-seconds 0 --steps 1 > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --workload powerlaw --spmv 6 --cpu-seconds 0 --steps 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --workload powerlaw --spmv 6 --cpu-seconds 0 --steps 1 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --workload powerlaw --spmv 6 --cpu-seconds 0 --steps 1 > $OUT/pmc_tcc.log 2>&1 || exit 1
find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/*/*kernel_stats.csv")
if f:
    for l in open(f[0]).read().splitlines()[:6]:
        print(l[:60], l.split('"')[-1] if '"' in l else "")
for tag in ("pmc_fetch", "pmc_write", "pmc_tcc"):
    f = glob.glob(out + "/%s/*/*counter_collection.csv" % tag)
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "binned" in r["Kernel_Name"] or "fill" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        v.sort()
        print(tag, k, "n", len(v), "median", v[len(v) // 2], "sum/spmv", sum(v) / (len(v) / 31.0) if "binned" in k[0] else "")
PY
