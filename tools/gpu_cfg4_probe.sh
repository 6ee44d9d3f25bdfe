#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 200 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
CO="lap2d:nx=3162,ny=3162,coef=1"
for rep in a b; do
for g in 0 512 768 1024 1536; do run coef_sg${g}_$rep "$CO" LSBENCH_HIP_SPMV_GRID=$g LSBENCH_HIP_SPMV_TUNE=6 LSBENCH_HIP_BLAS1_NT=41; done
done
run coef_auto "$CO" A=1
run cfg3_auto "lap2d:nx=3162,ny=3162" A=1
run cfg4_auto "lap3d:nx=400,ny=400,nz=400" A=1
