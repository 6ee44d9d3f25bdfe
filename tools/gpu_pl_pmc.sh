#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-plpmc}; mkdir -p $OUT; export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  n=$(echo $c | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$n" -- python3 tools/gpu_powerlaw.py 2000000 > $OUT/pmc_$n.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    # group consecutive dispatches: panel launches (9 args incl rowmap) vs plain
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"][:34] + " grid=" + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "spmv" in k or "k_dot" in k:
            print(k, " ".join("%s: total=%.3g n=%d" % (c, v, cnt[(k, c)]) for c, v in sorted(d.items())))
PY
