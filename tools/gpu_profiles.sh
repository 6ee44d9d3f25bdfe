#!/bin/bash
# Round profiles: the default bench, kernel-trace stats and PMC passes (their own runs,
# --kernel-trace only next to --pmc) for configs 3, 4, 5; config 2 and the optional
# forms as bench lines.  Condensed into profiles/ by tools/summarize_profiles.py.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-profiles}; mkdir -p $OUT; export TMPDIR=/tmp
# usage: gpu_profiles.sh <dir> [part]   part = A | B | C | D | E | all (a gpurun call is capped at 20 minutes: one part per call)
PART=${2:-all}; CUR=A
part() { CUR=$1; }
step() { local name=$1 secs=$2; shift 2
  if [ "$PART" != all ] && [ "$PART" != "$CUR" ]; then return 0; fi
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "rc=$rc"; if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --csr-kernel 0 --cfg2 0 --cfg5 0"
F=file:tests/golden/matrices/xn3b_A_18.txt.gz
step bench 600 python bench.py
# the driver's own command line (BENCH_rNN.json): 20 timed solves after 5 warm-ups, every sub-record
T0=$(date +%s); step bench_driver 600 python bench.py --gpus 1 --steps 20 --warmup 5; echo "bench_driver wall $(( $(date +%s) - T0 )) s"
step trace 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 1 --warmup 0 $Q
step pmc_fetch 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --fixed-iters 60 --steps 1 --warmup 0 $Q
step pmc_write 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --fixed-iters 60 --steps 1 --warmup 0 $Q
G="--workload lap2d_coef --general-values 0"
step bench_coef 400 python bench.py $G --steps 2 $Q
step trace_coef 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_coef" -- python3 bench.py $G --fixed-iters 200 --steps 1 --warmup 0 $Q
step pmc_fetch_coef 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_coef" -- python3 bench.py $G --fixed-iters 60 --steps 1 --warmup 0 $Q
step pmc_write_coef 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_coef" -- python3 bench.py $G --fixed-iters 60 --steps 1 --warmup 0 $Q
step bench_coef_fp32 400 python bench.py $G --precision fp32 --steps 2 $Q
step trace_coef_fp32 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_coef_fp32" -- python3 bench.py $G --precision fp32 --fixed-iters 200 --steps 1 --warmup 0 $Q
# SURVEY 8(d) to the letter: the kernels that stream 12 B per non-zero (bench.py's csr_kernel record)
step trace_csr 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_csr" -- python3 bench.py --only csr_kernel
step pmc_fetch_csr 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_csr" -- python3 bench.py --only csr_kernel
step pmc_write_csr 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_csr" -- python3 bench.py --only csr_kernel
# GMRES(30): throughput on config 3's pattern with general values (a fixed number of inner steps), and to
# the tolerance on the raw, unsymmetrised tests/xn3b_A_18.txt
step gmres_coef 400 python bench.py $G --krylov gmres --restart 30 --fixed-iters 300 --steps 2 --warmup 1 $Q
step trace_gmres 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_gmres" -- python3 bench.py $G --krylov gmres --restart 30 --fixed-iters 120 --steps 1 --warmup 0 $Q
step gmres_xn3b_raw 300 python bench.py --workload $F --operator raw --krylov gmres --restart 30 --tol 1e-10 --steps 20 --warmup 2 --verify 0 $Q
part B
step trace_lap3d 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_lap3d" -- python3 bench.py --workload lap3d --fixed-iters 100 --steps 1 --warmup 0 $Q
step pmc_fetch_lap3d 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_lap3d" -- python3 bench.py --workload lap3d --fixed-iters 40 --steps 1 --warmup 0 $Q
step pmc_write_lap3d 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_lap3d" -- python3 bench.py --workload lap3d --fixed-iters 40 --steps 1 --warmup 0 $Q
step bench_powerlaw 400 python bench.py --workload powerlaw $Q
step trace_powerlaw 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_powerlaw" -- python3 bench.py --workload powerlaw --steps 1 $Q
step pmc_fetch_powerlaw 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_powerlaw" -- python3 bench.py --workload powerlaw --steps 1 $Q
step pmc_write_powerlaw 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_powerlaw" -- python3 bench.py --workload powerlaw --steps 1 $Q
part C
for v in 7 6 1; do step bench_powerlaw_v$v 400 python bench.py --workload powerlaw --spmv $v $Q; done
step cfg5_spd_cg 600 python bench.py --workload "powerlaw:n=8000000,gamma=1.585350372615855,max=4096,seed=20240607,spd=1" --tol 1e-10 --steps 3 $Q
step trace_cfg2_fsai 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg2_fsai" -- python3 bench.py --workload $F --tol 1e-12 --steps 20 --warmup 2 --persistent 0 --precond fsai $Q
step trace_cfg2 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg2" -- python3 bench.py --workload $F --tol 1e-12 --steps 20 --warmup 2 --persistent 0 $Q
C2="--workload $F --tol 1e-12 --steps 100 --warmup 5 --cfg4 0 --verify 0"
step cfg2_launches 300 python bench.py $C2 --persistent 0
step cfg2_persistent 300 python bench.py $C2 --persistent 1 --cpu-seconds 0
step cfg2_dense_inverse 300 python bench.py $C2 --persistent 0 --precond bj --block-size 1000000 --cpu-seconds 0
step cfg2_cheb4 300 python bench.py $C2 --persistent 0 --precond cheb --cheb-degree 4 --cpu-seconds 0
step cfg2_fsai2 300 python bench.py $C2 --persistent 0 --precond fsai --fsai-power 2 --cpu-seconds 0
step cfg2_fsai3 300 python bench.py $C2 --persistent 0 --precond fsai --fsai-power 3 --cpu-seconds 0
LSBENCH_HIP_NO_FSAI_FUSE=1 step cfg2_fsai3_six_launches 300 python bench.py $C2 --persistent 0 --precond fsai --fsai-power 3 --cpu-seconds 0
LSBENCH_HIP_NO_TMPL=1 step cfg3_no_templates 400 python bench.py --steps 2 $Q
step cfg3_fp32 400 python bench.py --precision fp32 --steps 2 $Q
step cfg3_cheb4 400 python bench.py --precond cheb --cheb-degree 4 --steps 2 $Q
step cfg3_cheb16 400 python bench.py --precond cheb --cheb-degree 16 --steps 2 $Q
step cfg3_bj8 400 python bench.py --precond bj --block-size 8 --steps 2 $Q
find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
part D
# the PMC passes again with the SpMV flavour FORCED (spmv_tune flags; 5 = LSB_SPMV_SELL): the timing pass picks per
# box (and picks differently under the profiler), bench.py quotes the entry of the flavour its own run picked
pmc2() { local tag=$1; shift
  step pmc_fetch$tag 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch$tag" -- python3 bench.py "$@" --steps 1 --warmup 0 $Q
  step pmc_write$tag 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write$tag" -- python3 bench.py "$@" --steps 1 --warmup 0 $Q; }
pmc2 _f70 --fixed-iters 60 --spmv 5 --spmv-tune 70
pmc2 _f198 --fixed-iters 60 --spmv 5 --spmv-tune 198
pmc2 _lap3d_f326 --workload lap3d --fixed-iters 40 --spmv 5 --spmv-tune 326
pmc2 _lap3d_f198 --workload lap3d --fixed-iters 40 --spmv 5 --spmv-tune 198
pmc2 _lap3d_f70 --workload lap3d --fixed-iters 40 --spmv 5 --spmv-tune 70
export LSBENCH_HIP_FORCE_PERIOD=1
pmc2 _lap3d_f198p --workload lap3d --fixed-iters 40 --spmv 5 --spmv-tune 198
pmc2 _lap3d_f70p --workload lap3d --fixed-iters 40 --spmv 5 --spmv-tune 70
unset LSBENCH_HIP_FORCE_PERIOD
part E
pmc2 _coef_f6 --workload lap2d_coef --fixed-iters 60 --spmv 5 --spmv-tune 6
if [ -f $OUT/bench.log ]; then tail -c 400 $OUT/bench.log; fi
