#!/bin/bash
# config 4 / config 3: experiment switches of the template SpMV, one process each
set -e
out=gpurun_out/r3_cfg4probe5; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_sell.py tests/test_mixed.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 200 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
C4=lap3d:nx=400,ny=400,nz=400
C3=lap2d:nx=3162,ny=3162
run base "$C4" A=1
run col "$C4" LSBENCH_HIP_SELL_COL=1
run base2 "$C4" A=1
run col2 "$C4" LSBENCH_HIP_SELL_COL=1
run cfg3 "$C3" A=1
run cfg3b "$C3" A=1
run slab50 "lap3d:nx=400,ny=400,nz=50" A=1
