#!/bin/bash
# round 4: config 4's deferred-store template SpMV by grid size and dealing (the timing pass tries 1536 only)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_grid}; mkdir -p $out
run() { label=$1; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 100 lap3d:nx=400,ny=400,nz=400 >> $out/log.txt 2>$out/err_$label.txt; rc=$?; tail -1 $out/log.txt; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
for g in 768 1024 1280 1536 2048; do
  run c4_g${g}    PROBE_TUNE=198 PROBE_GRID=$g
  run c4_g${g}_p  PROBE_TUNE=198 PROBE_GRID=$g LSBENCH_HIP_FORCE_PERIOD=1
done
