#!/bin/bash
# One gpurun call: GPU test suite -> smoke -> bench -> rocprofv3 kernel trace.
# A step that is killed by its timeout ends the call (no further GPU step).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-round}
mkdir -p "$OUT"
step() { # name seconds cmd...
  local name=$1 secs=$2; shift 2
  echo "=== $name" | tee -a $OUT/steps.log
  timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/steps.log
  tail -n 6 "$OUT/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed: stopping"; exit $rc; fi
}
step pytest_gpu 900 python -m pytest tests -m gpu -x -q
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
step bench 400 python bench.py
export TMPDIR=/tmp
step rocprof_trace 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0
find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
ls -laR "$OUT/prof" | tail -n 12
step bench_cfg2 200 python bench.py --workload file:tests/golden/matrices/xn3b_A_18.txt.gz --tol 1e-12 --steps 200 --warmup 20
step bench_lap3d 400 python bench.py --workload lap3d --steps 2 --warmup 1 --cpu-seconds 0
step bench_powerlaw 400 python bench.py --workload powerlaw
