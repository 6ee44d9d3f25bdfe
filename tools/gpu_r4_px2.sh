#!/bin/bash
# round 4: k_pcg_col_px on config 4 -- nontemporal x / p' / q, fabric read requests of the launch
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_px2}; mkdir -p $OUT; export TMPDIR=/tmp
probe() { local label=$1 tune=$2 grid=$3 k=$4 spec=$5; shift 5
  ( export "$@" PROBE_TUNE=$tune PROBE_GRID=$grid LSBENCH_HIP_COL_K=$k PROBE_NOSAMPLE=1; timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 200 $spec >> $OUT/probe.txt 2>> $OUT/probe.err )
  local rc=$?; tail -n 1 $OUT/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
C4=lap3d:nx=400,ny=400,nz=400
SLAB=lap3d:nx=400,ny=400,nz=50
L2D=lap2d:nx=8192,ny=1220
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 400 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit $rc; fi; }
step pytest_px 600 python -m pytest tests/test_sell.py -m gpu -x -q -k "two_launch_column"
probe c4_three 326 1024 16 $C4 LSBENCH_HIP_NO_FUSE_PX=1
for nt in 2 3; do probe c4_two_nt$nt 326 1024 16 $C4 LSBENCH_HIP_PX_NT=$nt; done
probe c4_two_nt3_k8 326 1024 8 $C4 LSBENCH_HIP_PX_NT=3
for k in 4 6 12; do
probe slab_three_k$k 326 1024 $k $SLAB LSBENCH_HIP_NO_FUSE_PX=1
probe slab_two_k$k 326 1024 $k $SLAB A=1
probe l2d_three_k$k 326 1024 $k $L2D LSBENCH_HIP_NO_FUSE_PX=1
probe l2d_two_k$k 326 1024 $k $L2D A=1
done
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_HIT_sum"
( export PROBE_TUNE=326 PROBE_GRID=1024 LSBENCH_HIP_COL_K=16 PROBE_NOSAMPLE=1 LSBENCH_HIP_PX_NT=3; timeout -k 10 150 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/px_1 -- python3 tools/gpu_cfg4_probe.py px_1 40 $C4 >> $OUT/log.txt 2> $OUT/err_px_1.txt ); rc=$?
find $OUT/px_1 -name '*kernel_trace.csv' -delete 2>/dev/null
if [ $rc -ge 124 ]; then echo "pmc killed"; exit $rc; fi
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
grep -B1 -A8 "k_pcg_col_px\|k_pcg_update_r" $OUT/summary.txt
