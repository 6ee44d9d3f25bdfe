#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-lab5}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 tools/spmv_lab powerlaw 2000000 7 > $OUT/lab.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
grep -vE "^check.*ok" $OUT/lab.log
for c in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -c1-6)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$n" -- tools/spmv_lab powerlaw 2000000 1 > $OUT/pmc_$n.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "cyc<2048, 1" in k or "k_adaptive<2048, 0>" in k:
            print(k[5:44], " ".join("%s=%.3g" % (c.replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(d.items())))
PY
