#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pl_blocks}; mkdir -p $OUT; export TMPDIR=/tmp
for K in ${2:-1 8 16 32}; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$K -- python3 tools/gpu_pl_blocks.py $K > $OUT/k$K.log 2>&1 || { echo "K=$K failed"; tail -5 $OUT/k$K.log; exit 1; }
  find $OUT -name '*kernel_trace.csv' -delete
  python3 - $OUT/k$K $K <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
K = int(sys.argv[2]); tot = 0.0
for r in csv.DictReader(open(f)):
    if "k_pb_" in r["Name"]:
        per = float(r["TotalDurationNs"]) / 12 / 1e3
        tot += per
        print("  K=%d %-16s %5d calls, %.1f us per SpMV" % (K, r["Name"].split("(")[0][-16:], int(r["Calls"]), per))
print("K=%d: %.1f us of two-phase kernels per SpMV" % (K, tot))
PY
done
