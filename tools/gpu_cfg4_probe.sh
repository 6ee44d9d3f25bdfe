#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_sell.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -1 $out/tests.log
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 200 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
C4=lap3d:nx=400,ny=400,nz=400
C3=lap2d:nx=3162,ny=3162
for rep in a b; do
run c4_t70_$rep "$C4" PROBE_TUNE=70 PROBE_GRID=1536 LSBENCH_HIP_BLAS1_NT=41
run c4_t198_$rep "$C4" PROBE_TUNE=198 PROBE_GRID=1536 LSBENCH_HIP_BLAS1_NT=41
run c4_t198p_$rep "$C4" PROBE_TUNE=198 PROBE_GRID=1536 LSBENCH_HIP_FORCE_PERIOD=1 LSBENCH_HIP_BLAS1_NT=41
run c3_t70_$rep "$C3" PROBE_TUNE=70 PROBE_GRID=1536 LSBENCH_HIP_BLAS1_NT=41
run c3_t198_$rep "$C3" PROBE_TUNE=198 PROBE_GRID=1536 LSBENCH_HIP_BLAS1_NT=41
done
