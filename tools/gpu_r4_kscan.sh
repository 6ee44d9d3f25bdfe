#!/bin/bash
# round 4: the two-launch iteration by column length (LSBENCH_HIP_COL_K) and by the grid its launches inherit
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_kscan}; mkdir -p $OUT
probe() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 400 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then exit $rc; fi; }
C3=lap2d:nx=3162,ny=3162
if [ "${2:-k}" = grid ]; then
  for g in 768 1024 1280 1536; do probe c3_g$g $C3 PROBE_TUNE=326 PROBE_GRID=$g; done
  for g in 768 1024 1280 1536; do probe c3_k4_g$g $C3 PROBE_TUNE=326 PROBE_GRID=$g LSBENCH_HIP_COL_K=4; done
else
  for k in 3 4 5 6 8 12; do probe c3_k$k $C3 LSBENCH_HIP_COL_K=$k; done
  for k in 8 12 16; do probe c4_k$k lap3d:nx=400,ny=400,nz=400 LSBENCH_HIP_COL_K=$k; done
fi
