#!/bin/bash
# round 4, second session: the z-column walk of the template layout (k_spmv_tmpl_col) -- parity test, then config 4
# back to back and in the solve against the deferred-store flavour, by column length and grid
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_col}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 700 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
step pytest_col 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "z_column_walk"
probe() { local label=$1 tune=$2 grid=$3 k=$4
  PROBE_TUNE=$tune PROBE_GRID=$grid LSBENCH_HIP_COL_K=$k timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 200 >> $OUT/probe.txt 2>> $OUT/probe.err
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
probe c4_defer 198 1536 8
probe c4_col_k8 326 1536 8
probe c4_col_k16 326 1536 16
probe c4_col_k4 326 1536 4
probe c4_col_k8_g1024 326 1024 8
probe c4_col_k8_g2048 326 2048 8
