#!/bin/bash
# config 5, two-phase SpMV cut into K row super-blocks (K virtual shards): do a block's products stay
# in the 256 MB Infinity Cache between its two phases when they are stored with PLAIN stores
# (LSBENCH_HIP_PB_NTSTORE=0) instead of nontemporal ones?  Summed kernel time per SpMV from rocprofv3.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pl_blocks2}; mkdir -p $OUT; export TMPDIR=/tmp
for K in ${KS:-1 8 16 32}; do for nt in 1 0; do
  d=$OUT/k${K}_nt$nt
  LSBENCH_HIP_PB_NTSTORE=$nt timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/gpu_pl_blocks.py $K > $d.log 2>&1; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
  python3 - $d $K $nt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
K = int(sys.argv[2]); tot = {}
for r in csv.DictReader(open(f)):
    for k in ("k_pb_products", "k_pb_reduce"):
        if k in r["Name"]:
            tot[k] = (float(r["TotalDurationNs"]), int(r["Calls"]))
spmvs = tot["k_pb_products"][1] / K
print("K=%2d ntstore=%s: phase 1 %.0f us + phase 2 %.0f us = %.0f us per SpMV (%d SpMVs)" % (
    K, sys.argv[3], tot["k_pb_products"][0] / spmvs / 1e3, tot["k_pb_reduce"][0] / spmvs / 1e3,
    (tot["k_pb_products"][0] + tot["k_pb_reduce"][0]) / spmvs / 1e3, spmvs), flush=True)
PY
  rm -rf $d
done; done
