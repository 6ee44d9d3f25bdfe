#!/bin/bash
# round 4: what bounds config 4's template SpMV?  Fabric-side request counts, queue LEVELs (-> mean latency),
# DRAM share, stalls and SQ wait buckets of the launch under the two dealings, next to the 64 M-row 5-point
# operator and the BLAS-1 sweeps of the same solve (which stream the same vectors at ~6 TB/s).  Counters in
# their own passes (--pmc with --kernel-trace only).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_cfg4_pmc}; mkdir -p $out; export TMPDIR=/tmp
C4=lap3d:nx=400,ny=400,nz=400
C5=lap2d:nx=8000,ny=8000
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum"
P2="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_MISS_sum"
P3="TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
P4="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run() { label=$1; spec=$2; pass=$3; ctrs=$4; shift 4
  ( export "$@" PROBE_GRID=1536; timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/${label}_$pass -- python3 tools/gpu_cfg4_probe.py ${label}_$pass 40 "$spec" >> $out/log.txt 2> $out/err_${label}_$pass.txt ); rc=$?
  tail -1 $out/log.txt; find $out/${label}_$pass -name '*kernel_trace.csv' -delete 2>/dev/null
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
for pass in 1 2 3 4; do
  eval ctrs=\$P$pass
  run contig "$C4" $pass "$ctrs" PROBE_TUNE=198
  run period "$C4" $pass "$ctrs" PROBE_TUNE=198 LSBENCH_HIP_FORCE_PERIOD=1
  run five   "$C5" $pass "$ctrs" PROBE_TUNE=70
done
python3 tools/summarize_pmc.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
