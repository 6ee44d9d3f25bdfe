#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 200 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
run minw7 "lap3d:nx=400,ny=400,nz=400" A=1
run minw7b "lap3d:nx=400,ny=400,nz=400" A=1
