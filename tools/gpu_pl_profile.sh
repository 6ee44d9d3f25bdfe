#!/bin/bash
# Config 5 (power law, 8 M rows) evidence: kernel-trace stats + PMC traffic of the
# bench's SpMV launches (separate passes, kernel-trace only next to --pmc).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-plprof}; mkdir -p $OUT; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2
  echo "=== $name"; timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "rc=$rc"; if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi; }
ARGS="--workload powerlaw --steps 2 --warmup 1 --cpu-seconds 0 ${PL_ARGS:-}"
step bench 500 python3 bench.py $ARGS
step trace 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS
step pmc_fetch 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS
step pmc_write 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS
step pmc_tcc 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -- python3 bench.py $ARGS
find "$OUT" -name '*kernel_trace.csv' -size +12M -delete 2>/dev/null
tail -n 2 $OUT/bench.log
