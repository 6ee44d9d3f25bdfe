#!/bin/bash
# Config 5, two-phase SpMV (hip_pb.hip): per-phase kernel times by tiling (columns per
# chunk x rows per bin), against the binned form.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-pl_twophase}; mkdir -p $OUT; export TMPDIR=/tmp
Q="--workload powerlaw --steps 1 --cpu-seconds 0 --cfg4 0 --verify 0"
for t in ${2:-16384x1024 16384x2048 8192x1024 8192x2048}; do
  c=${t%x*}; r=${t#*x}
  LSBENCH_HIP_PB_COLS=$c LSBENCH_HIP_PB_ROWS=$r timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OUT/t_$t -- python3 bench.py $Q --spmv 7 > $OUT/t_$t.log 2>&1 || { echo "tiling $t failed"; tail -5 $OUT/t_$t.log; exit 1; }
  echo "tiling $t: $(grep -o '"launch_ms": [0-9.]*' $OUT/t_$t.log | head -1)"
  grep -h "k_pb_" $OUT/t_$t/*/*kernel_stats.csv | awk -F'","' '{printf "   %s calls %s avg %.1f us\n", substr($1,2,40), $2, $4/1000}'
  find $OUT -name '*kernel_trace.csv' -delete
done
