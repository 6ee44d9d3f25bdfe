#!/bin/bash
# round 4: z-column SpMV -- parity (incl. the interior plan of split launches), lockstep barrier A/B, a slab that fits
# the Infinity Cache (is the launch HBM-bound?), and what the timing pass picks on config 4
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_col2}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 700 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
step pytest_col 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "z_column_walk or plane_periodic"
probe() { local label=$1 tune=$2 grid=$3 k=$4 spec=$5; shift 5
  ( export "$@" PROBE_TUNE=$tune PROBE_GRID=$grid LSBENCH_HIP_COL_K=$k; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 200 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C4=lap3d:nx=400,ny=400,nz=400
SLAB=lap3d:nx=400,ny=400,nz=50
probe c4_sync_k8_g1024 326 1024 8 $C4 A=1
probe c4_nosync_k8_g1024 326 1024 8 $C4 LSBENCH_HIP_COL_NOSYNC=1
probe c4_sync_k8_g768 326 768 8 $C4 A=1
probe c4_sync_k8_g1536 326 1536 8 $C4 A=1
probe c4_sync_k16_g1024 326 1024 16 $C4 A=1
probe slab_defer 198 1536 8 $SLAB A=1
probe slab_col_nosync 326 1024 8 $SLAB LSBENCH_HIP_COL_NOSYNC=1
probe slab_col_sync 326 1024 8 $SLAB A=1
probe slab_col_sync_k16 326 1024 16 $SLAB A=1
( unset PROBE_TUNE; timeout -k 10 300 python tools/gpu_cfg4_probe.py c4_auto 200 $C4 >> $OUT/probe.txt 2>> $OUT/probe.err ); tail -n 1 $OUT/probe.txt
