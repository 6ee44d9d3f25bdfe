#!/bin/bash
# kernel-trace stats of config 3 (fixed 400 iterations): per-kernel means, A/B of LSBENCH_HIP_UPD_P_X2
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_trace}; mkdir -p $OUT; export TMPDIR=/tmp
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --fixed-iters 400 --steps 2 --warmup 0"
for x2 in 1 0; do
  export LSBENCH_HIP_UPD_P_X2=$x2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/x2_$x2" -- python3 bench.py $Q ${@:2} > $OUT/x2_$x2.log 2>&1; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
  f=$(ls $OUT/x2_$x2/*/*kernel_stats.csv | head -n 1)
  echo "== X2=$x2"; cut -d, -f1-4 $f | head -n 8 | cut -c1-150
  find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
done
grep -o '"launch_ms": [0-9.]*' $OUT/x2_1.log | head -n 2
