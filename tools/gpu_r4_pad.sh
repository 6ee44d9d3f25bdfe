#!/bin/bash
# round 4: line padding of the 2-D grid (config 3: lines of 3162 rows -> 3200): parity, then the headline iteration padded / unpadded
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_pad}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 1500 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit $rc; fi; }
step pytest_pad 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "line_padded or two_launch_column"
probe() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 400 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C3=lap2d:nx=3162,ny=3162
probe c3_unpadded $C3 LSBENCH_HIP_PAD_LINES=0
probe c3_padded_three $C3 LSBENCH_HIP_NO_FUSE_PX=1
probe c3_padded_two $C3 A=1
probe c3_padded_two_k4 $C3 LSBENCH_HIP_COL_K=4
probe c3_padded_two_k8 $C3 LSBENCH_HIP_COL_K=8
probe c3_padded_two_326 $C3 PROBE_TUNE=326 PROBE_GRID=1024
