#!/bin/bash
set -e
out=gpurun_out/${1:-r3_probe}; mkdir -p $out
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 100 "$spec" >> $out/log.txt 2>$out/err_$label.txt; tail -1 $out/log.txt; }
C4=lap3d:nx=400,ny=400,nz=400
run c4_forced "$C4" PROBE_TUNE=70 PROBE_GRID=1536 LSBENCH_HIP_FORCE_PERIOD=1
run c4_nostore "$C4" PROBE_TUNE=70 PROBE_GRID=1536 LSBENCH_HIP_FORCE_PERIOD=1 LSBENCH_HIP_TMPL_NOSTORE=1
run c4_nostore_ahead "$C4" PROBE_TUNE=70 LSBENCH_HIP_FORCE_PERIOD=1 LSBENCH_HIP_TMPL_NOSTORE=1 LSBENCH_HIP_TMPL_AHEAD=1
