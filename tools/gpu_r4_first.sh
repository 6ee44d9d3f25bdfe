#!/bin/bash
# round 4, first call: the new GPU tests, the default bench line (all five configs now), GMRES / fp32 lines
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r4_first}; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"; local rc=$?
  echo "rc=$rc"; tail -c 600 "$OUT/$name.log"; tail -n 3 "$OUT/$name.err"
  if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
T0=$(date +%s); timeout -k 10 600 python bench.py --verbose 2 > $OUT/bench.log 2> $OUT/bench.err; rc=$?
echo "bench rc=$rc wall=$(( $(date +%s) - T0 )) s"; tail -c 3000 $OUT/bench.log; tail -n 5 $OUT/bench.err
if [ $rc -ge 124 ]; then exit $rc; fi
Q="--cpu-seconds 0 --cfg4 0 --general-values 0 --csr-kernel 0 --cfg2 0 --cfg5 0"
G="--workload lap2d_coef"
F=file:tests/golden/matrices/xn3b_A_18.txt.gz
step gmres_coef 400 python bench.py $G --krylov gmres --restart 30 --fixed-iters 300 --steps 2 --warmup 1 $Q
step gmres_xn3b_raw 300 python bench.py --workload $F --operator raw --krylov gmres --restart 30 --tol 1e-10 --steps 20 --warmup 2 --verify 0 $Q
step bench_coef_fp32 400 python bench.py $G --precision fp32 --steps 2 $Q
step bench_coef 400 python bench.py $G --steps 2 $Q
step pytest_contract 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bench_line_contract"
