#!/bin/bash
# config 5, two-phase SpMV: phase 1's loads of trip t+1 issued before (1) / behind (3) the stores of trip t, same box
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pl_pipe}; mkdir -p $OUT
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --workload powerlaw --spmv 7"
for rep in 1 2 3; do
for cfg in "1 8192" "3 8192" "1 32768" "3 32768"; do
  set -- $cfg
  LSBENCH_HIP_PB_NTSTORE=$1 LSBENCH_HIP_PB_ITEM=$2 timeout -k 10 300 python bench.py $Q > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  python3 - "$1" "$2" $OUT/run.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][0])
print("order %s item %6s: SpMV %.1f us = %.1f GB/s" % ("ahead " if sys.argv[1] == "1" else "behind", sys.argv[2], d["ms_per_step"] * 1e3, d["value"]), flush=True)
PY
done
done
