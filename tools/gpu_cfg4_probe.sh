#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 200 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
C4=lap3d:nx=400,ny=400,nz=400
C3=lap2d:nx=3162,ny=3162
for rep in a b c; do
run c4_new_$rep "$C4" LSBENCH_HIP_BLAS1_NT=41
run c4_old_$rep "$C4" LSBENCH_HIP_BLAS1_NT=41 LSBENCH_HIP_XR_OLD=1
run c3_new_$rep "$C3" LSBENCH_HIP_BLAS1_NT=41
run c3_old_$rep "$C3" LSBENCH_HIP_BLAS1_NT=41 LSBENCH_HIP_XR_OLD=1
done
