#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 100 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
C4=lap3d:nx=400,ny=400,nz=400
for g in 1024 1280 1536 1792 2048; do
run c4_d_g$g "$C4" PROBE_TUNE=198 PROBE_GRID=$g LSBENCH_HIP_BLAS1_NT=41
done
for g in 1280 1536 2048; do
run c4_dp_g$g "$C4" PROBE_TUNE=198 PROBE_GRID=$g LSBENCH_HIP_FORCE_PERIOD=1 LSBENCH_HIP_BLAS1_NT=41
done
