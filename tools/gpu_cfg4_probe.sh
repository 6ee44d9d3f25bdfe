#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 100 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
B="lap2d:nx=8000,ny=8000"
for rep in a b; do
run big2d_t70_$rep "$B" PROBE_TUNE=70 PROBE_GRID=1536 LSBENCH_HIP_BLAS1_NT=41
run big2d_t198_$rep "$B" PROBE_TUNE=198 PROBE_GRID=1536 LSBENCH_HIP_BLAS1_NT=41
done
run big2d_auto "$B" A=1
