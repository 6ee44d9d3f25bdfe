#!/bin/bash
# round 4: z-column plan dealt to the XCDs in equal contiguous runs: parity, the padded config 3, config 4, the slab
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_pad3}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 600 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit $rc; fi; }
step pytest_pad 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "line_padded or two_launch_column or z_column_walk"
probe() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 400 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C3=lap2d:nx=3162,ny=3162
C4=lap3d:nx=400,ny=400,nz=400
probe c3_unpadded $C3 LSBENCH_HIP_PAD_LINES=0
probe c3_padded_auto $C3 A=1
probe c3_padded_two_326 $C3 PROBE_TUNE=326 PROBE_GRID=1024
probe c3_padded_two_326_k6 $C3 PROBE_TUNE=326 PROBE_GRID=1024 LSBENCH_HIP_COL_K=6
probe c3_padded_three_326 $C3 PROBE_TUNE=326 PROBE_GRID=1024 LSBENCH_HIP_NO_FUSE_PX=1
probe l2d_two lap2d:nx=8192,ny=1220 PROBE_TUNE=326 PROBE_GRID=1024
probe c4_auto $C4 A=1
probe c4_three $C4 LSBENCH_HIP_NO_FUSE_PX=1
probe slab_auto lap3d:nx=400,ny=400,nz=50 A=1
