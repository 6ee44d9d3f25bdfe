#!/bin/bash
# Round profiles: kernel-trace stats of the default bench + PMC passes (own runs).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-profiles}; mkdir -p $OUT; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "rc=$rc"; if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
step bench 400 python bench.py
step trace 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0
step pmc_fetch 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --fixed-iters 60 --steps 1 --warmup 0 --cpu-seconds 0
step pmc_write 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --fixed-iters 60 --steps 1 --warmup 0 --cpu-seconds 0
step trace_lap3d 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_lap3d" -- python3 bench.py --workload lap3d --fixed-iters 100 --steps 1 --warmup 0 --cpu-seconds 0
step trace_cfg2 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg2" -- python3 bench.py --workload file:tests/golden/matrices/xn3b_A_18.txt.gz --tol 1e-12 --steps 20 --warmup 2 --cpu-seconds 0
find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
tail -n 3 $OUT/bench.log
