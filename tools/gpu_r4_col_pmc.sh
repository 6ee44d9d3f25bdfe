#!/bin/bash
# round 4: fabric traffic and wait buckets of the z-column SpMV on config 4 (own --pmc passes), next to the deferred-store flavour
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_col_pmc}; mkdir -p $out; export TMPDIR=/tmp
C4=lap3d:nx=400,ny=400,nz=400
P0="FETCH_SIZE"
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum"
P2="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_MISS_sum"
P4="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
run() { label=$1; spec=$2; pass=$3; ctrs=$4; shift 4
  ( export "$@"; timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/${label}_$pass -- python3 tools/gpu_cfg4_probe.py ${label}_$pass 40 "$spec" >> $out/log.txt 2> $out/err_${label}_$pass.txt ); rc=$?
  tail -1 $out/log.txt; find $out/${label}_$pass -name '*kernel_trace.csv' -delete 2>/dev/null
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
for pass in 0 1 2 4; do
  eval ctrs=\$P$pass
  run col8 "$C4" $pass "$ctrs" PROBE_TUNE=326 PROBE_GRID=1024 LSBENCH_HIP_COL_K=8
done
python3 tools/summarize_pmc.py $out > $out/summary.txt 2>&1
grep -A14 "k_spmv_tmpl" $out/summary.txt
probe() { local label=$1 tune=$2 grid=$3 k=$4
  PROBE_TUNE=$tune PROBE_GRID=$grid LSBENCH_HIP_COL_K=$k timeout -k 10 240 python tools/gpu_cfg4_probe.py $label 200 >> $out/probe.txt 2>> $out/probe.err
  local rc=$?; tail -n 1 $out/probe.txt; if [ $rc -ge 124 ]; then echo "probe $label killed: stopping"; exit $rc; fi; }
probe c4_col_k8_g768 326 768 8
probe c4_col_k8_g896 326 896 8
probe c4_col_k8_g1024 326 1024 8
probe c4_col_k8_g1280 326 1280 8
probe c4_col_k16_g1024 326 1024 16
probe c4_col_k12_g1024 326 1024 12
