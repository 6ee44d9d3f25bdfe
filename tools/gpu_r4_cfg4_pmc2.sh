#!/bin/bash
# round 4: instruction mix of the template SpMV per launch (is the launch bound by instruction issue?)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r4_cfg4_pmc2}; mkdir -p $out; export TMPDIR=/tmp
C4=lap3d:nx=400,ny=400,nz=400
C5=lap2d:nx=8000,ny=8000
C3=lap2d:nx=3162,ny=3162
P5="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
P6="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64"
run() { label=$1; spec=$2; pass=$3; ctrs=$4; shift 4
  ( export "$@" PROBE_GRID=1536; timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/${label}_$pass -- python3 tools/gpu_cfg4_probe.py ${label}_$pass 40 "$spec" >> $out/log.txt 2> $out/err_${label}_$pass.txt ); rc=$?
  tail -1 $out/log.txt; find $out/${label}_$pass -name '*kernel_trace.csv' -delete 2>/dev/null
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; }
for pass in 5 6; do
  eval ctrs=\$P$pass
  run contig "$C4" $pass "$ctrs" PROBE_TUNE=198
  run plain  "$C4" $pass "$ctrs" PROBE_TUNE=70
  run five   "$C5" $pass "$ctrs" PROBE_TUNE=70
  run cfg3   "$C3" $pass "$ctrs" PROBE_TUNE=70
done
python3 tools/summarize_pmc.py $out > $out/summary.txt 2>&1
grep -A12 "k_spmv_tmpl" $out/summary.txt
