#!/bin/bash
# A/B on ONE box: BLAS-1 nontemporal loads+stores on/off, per-kernel times from rocprofv3.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/ab_nt; mkdir -p $OUT; export TMPDIR=/tmp
for nt in 1 0; do
  export LSBENCH_HIP_BLAS1_NT=$nt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nt$nt -- python3 bench.py --fixed-iters 400 --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/nt$nt.log 2>&1 || exit 1
  f=$(ls $OUT/nt$nt/*/*kernel_stats.csv | tail -1)
  echo "== BLAS1_NT=$nt"; head -4 $f | cut -c1-60,200-400 | sed 's/([^)]*)//'
  grep -o '"ms_per_step": [0-9.]*' $OUT/nt$nt.log
  find $OUT -name '*kernel_trace.csv' -delete
done
