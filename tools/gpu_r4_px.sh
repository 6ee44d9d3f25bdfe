#!/bin/bash
# round 4: the two-launch PCG iteration on a z-column plan (k_pcg_col_px + k_pcg_update_r): parity, then config 4's iteration
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_px}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 1500 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit $rc; fi; }
step pytest_px 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "two_launch_column or z_column_walk"
probe() { local label=$1 tune=$2 grid=$3 k=$4 spec=$5; shift 5
  ( export "$@" PROBE_TUNE=$tune PROBE_GRID=$grid LSBENCH_HIP_COL_K=$k; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 200 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C4=lap3d:nx=400,ny=400,nz=400
SLAB=lap3d:nx=400,ny=400,nz=50
probe c4_three_launches 326 1024 16 $C4 LSBENCH_HIP_NO_FUSE_PX=1 PROBE_NOSAMPLE=1
probe c4_two_launches 326 1024 16 $C4 PROBE_NOSAMPLE=1
probe c4_two_launches_g768 326 768 16 $C4 PROBE_NOSAMPLE=1
probe c4_two_launches_k8 326 1024 8 $C4 PROBE_NOSAMPLE=1
probe slab_three 326 1024 16 $SLAB LSBENCH_HIP_NO_FUSE_PX=1 PROBE_NOSAMPLE=1
probe slab_two 326 1024 16 $SLAB PROBE_NOSAMPLE=1
