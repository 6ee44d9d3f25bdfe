#!/bin/bash
# round 4: the template SpMV with 2 / 4 turns of a wave in flight (LSB_SP_DEEP2 / DEEP4) on config 4
# (and config 3), against the shipped flavours; bit-identity first
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_deep}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "constant_slots or repeat_under" > $out/pytest.log 2>&1; rc=$?
tail -n 4 $out/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() { label=$1; spec=$2; shift; shift; env "$@" timeout -k 10 300 python tools/gpu_cfg4_probe.py "$label" 100 "$spec" >> $out/log.txt 2>$out/err_$label.txt; rc=$?; tail -1 $out/log.txt; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
C4=lap3d:nx=400,ny=400,nz=400
C3=lap2d:nx=3162,ny=3162
run c4_tmpl        "$C4" PROBE_TUNE=70  PROBE_GRID=1536
run c4_defer       "$C4" PROBE_TUNE=198 PROBE_GRID=1536
run c4_defer_per   "$C4" PROBE_TUNE=198 PROBE_GRID=1536 LSBENCH_HIP_FORCE_PERIOD=1
run c4_deep2_1280  "$C4" PROBE_TUNE=326 PROBE_GRID=1280
run c4_deep2_1536  "$C4" PROBE_TUNE=326 PROBE_GRID=1536
run c4_deep2_per   "$C4" PROBE_TUNE=326 PROBE_GRID=1280 LSBENCH_HIP_FORCE_PERIOD=1
run c4_deep4_768   "$C4" PROBE_TUNE=582 PROBE_GRID=768
run c4_deep4_1536  "$C4" PROBE_TUNE=582 PROBE_GRID=1536
run c4_deep4_per   "$C4" PROBE_TUNE=582 PROBE_GRID=768 LSBENCH_HIP_FORCE_PERIOD=1
run c3_tmpl        "$C3" PROBE_TUNE=70  PROBE_GRID=1536
run c3_deep2       "$C3" PROBE_TUNE=326 PROBE_GRID=1280
run c3_deep4       "$C3" PROBE_TUNE=582 PROBE_GRID=768
cat $out/log.txt
