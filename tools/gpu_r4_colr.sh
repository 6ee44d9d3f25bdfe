#!/bin/bash
# round 4: the r half of the iteration with S p formed again (k_pcg_col_r; q not stored): parity, configs 3 and 4, the slab
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_colr}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 600 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit $rc; fi; }
step pytest 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "line_padded or two_launch_column or z_column_walk"
probe() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 400 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C3=lap2d:nx=3162,ny=3162
C4=lap3d:nx=400,ny=400,nz=400
probe c3_two $C3 A=1
probe c3_three $C3 LSBENCH_HIP_NO_FUSE_PX=1
probe c4_two $C4 A=1
probe c4_three $C4 LSBENCH_HIP_NO_FUSE_PX=1
probe slab_two lap3d:nx=400,ny=400,nz=50 A=1
probe slab_three lap3d:nx=400,ny=400,nz=50 LSBENCH_HIP_NO_FUSE_PX=1
