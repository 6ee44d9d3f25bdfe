#!/bin/bash
# binned SpMV on config 5: gather flavours x window widths (bench.py --spmv 6 --spmv-tune F)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-plf}; mkdir -p $OUT
for w in 262144 524288 1048576; do for f in 2 10 18 0 8; do
  LSBENCH_HIP_PANEL_COLS=$w timeout -k 10 300 python3 bench.py --workload powerlaw --spmv 6 --spmv-tune $f --cpu-seconds 0 > $OUT/w${w}_f$f.log 2>&1 || exit 1
  echo "width $w flags $f: $(grep -o '"ms_per_step": [0-9.]*' $OUT/w${w}_f$f.log)"
done; done
