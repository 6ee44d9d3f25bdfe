#!/bin/bash
# round 4: k_pcg_col_px with results stored right away (96 / 142 VGPRs): parity, config 3 padded by grid, config 4
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_px5}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 300 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit $rc; fi; }
step pytest_px 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "line_padded or two_launch_column"
probe() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 400 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C3=lap2d:nx=3162,ny=3162
for g in 768 1024 1280; do probe c3_two_g$g $C3 PROBE_TUNE=326 PROBE_GRID=$g; done
probe c3_two_g1280_k6 $C3 PROBE_TUNE=326 PROBE_GRID=1280 LSBENCH_HIP_COL_K=6
probe c3_three $C3 PROBE_TUNE=326 PROBE_GRID=1024 LSBENCH_HIP_NO_FUSE_PX=1
probe c4_two lap3d:nx=400,ny=400,nz=400 PROBE_TUNE=326 PROBE_GRID=768
probe c4_three lap3d:nx=400,ny=400,nz=400 PROBE_TUNE=326 PROBE_GRID=1024 LSBENCH_HIP_NO_FUSE_PX=1
