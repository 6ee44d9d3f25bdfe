#!/bin/bash
# round 4: k_pcg_col_px on config 4 by grid (fewer resident workgroups: more of the +-line gathers out of L2?), smoke
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_px4}; mkdir -p $OUT
probe() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 300 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C4=lap3d:nx=400,ny=400,nz=400
for g in 384 512 640 768; do probe c4_two_g$g $C4 PROBE_TUNE=326 PROBE_GRID=$g; done
probe c4_two_g768_k8 $C4 PROBE_TUNE=326 PROBE_GRID=768 LSBENCH_HIP_COL_K=8
probe c4_two_g512_k8 $C4 PROBE_TUNE=326 PROBE_GRID=512 LSBENCH_HIP_COL_K=8
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
