#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 300 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
C4=lap3d:nx=400,ny=400,nz=400
C3=lap2d:nx=3162,ny=3162
CO="lap2d:nx=3162,ny=3162,coef=1"
for rep in a b; do
run c3_classic_$rep "$C3" A=1
run c3_cg1_$rep "$C3" PROBE_KRYLOV=cg1
run c3_cg1_nt9_$rep "$C3" PROBE_KRYLOV=cg1 LSBENCH_HIP_BLAS1_NT=9
run c4_cg1_$rep "$C4" PROBE_KRYLOV=cg1
run co_classic_$rep "$CO" A=1
run co_cg1_$rep "$CO" PROBE_KRYLOV=cg1
run slab_classic_$rep "lap3d:nx=400,ny=400,nz=50" A=1
run slab_cg1_$rep "lap3d:nx=400,ny=400,nz=50" PROBE_KRYLOV=cg1
done
