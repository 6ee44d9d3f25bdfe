#!/bin/bash
# every nontemporal mask of the classic sweeps (bits 0-4) on one operator, one process each
set -e
out=gpurun_out/${1:-nt_scan}; mkdir -p $out
spec=${2:-lap2d:nx=3162,ny=3162}
for m in $(seq 0 31); do
  LSBENCH_HIP_BLAS1_NT=$((m == 1 ? 33 : m)) timeout -k 10 300 python tools/gpu_cfg4_probe.py "mask_$m" 300 "$spec" >> $out/log.txt 2>$out/err.txt
  tail -1 $out/log.txt | cut -c1-150
done
