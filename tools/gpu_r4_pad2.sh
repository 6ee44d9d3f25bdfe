#!/bin/bash
# round 4: kernel times of the two-launch iteration on the padded config 3 against the 8192-wide grid
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_pad2}; mkdir -p $OUT; export TMPDIR=/tmp
run() { local label=$1 spec=$2; shift 2
  ( export "$@" PROBE_NOSAMPLE=1 PROBE_TUNE=326 PROBE_GRID=1024; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$label -- python3 tools/gpu_cfg4_probe.py $label 300 $spec >> $OUT/log.txt 2> $OUT/err_$label.txt ); rc=$?
  tail -1 $OUT/log.txt; find $OUT/$label -name '*kernel_trace.csv' -delete 2>/dev/null
  f=$(find $OUT/$label -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
run c3pad lap2d:nx=3162,ny=3162 A=1
run l2d lap2d:nx=8192,ny=1220 A=1
run c3pad_k8 lap2d:nx=3162,ny=3162 LSBENCH_HIP_COL_K=8
